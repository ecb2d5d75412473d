// abi_asan_check.cpp -- the argument-validation paths of the C ABI (include/hoqp.h) that run WITHOUT a GPU, under
// ASan + UBSan (host code only: `make -C tests -f Makefile.asan`, or tests/test_abi.py::test_abi_under_sanitizers).
// Compiled together with the library's host side (hipcc --cuda-host-only -fsanitize=address,undefined), so every
// byte the entry points touch before they reach the device is instrumented.  With a GPU present the same
// program also creates and destroys a handle.
#include <cstdio>
#include <cstring>

#include "../include/hoqp.h"

#define EXPECT(cond)                                                     \
	do {                                                             \
		if (!(cond)) {                                           \
			std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); \
			++bad;                                           \
		}                                                        \
	} while (0)

int main()
{
	int bad = 0;
	hoqp_config cfg;
	hoqp_default_config(&cfg);
	EXPECT(cfg.eps == 1e-12 && cfg.mu == 0.5 && cfg.tau_max == 100.0 && cfg.semantics == HOQP_SEMANTICS_REFERENCE);
	hoqp_handle *h = nullptr;
	EXPECT(hoqp_create(nullptr, &h) == -1);
	EXPECT(hoqp_create(&cfg, nullptr) == -1);
	cfg.device = -1;
	EXPECT(hoqp_create(&cfg, &h) != 0 && h == nullptr);
	cfg.device = 1 << 20;
	EXPECT(hoqp_create(&cfg, &h) != 0 && h == nullptr);
	cfg.device = 0;
	EXPECT(hoqp_destroy(nullptr) == -1);
	EXPECT(std::strcmp(hoqp_last_error_string(nullptr), "null handle") == 0);
	// every device entry point refuses a null handle before it looks at anything else
	hoqp_task_batch tb;
	hoqp_result_batch rb;
	hoqp_wbc_inputs wi;
	hoqp_wbc_outputs wo;
	std::memset(&tb, 0, sizeof(tb));
	std::memset(&rb, 0, sizeof(rb));
	std::memset(&wi, 0, sizeof(wi));
	std::memset(&wo, 0, sizeof(wo));
	EXPECT(hoqp_solve_tasks_batched(nullptr, &tb, &rb, nullptr) == -1);
	EXPECT(hoqp_solve_tasks_host(nullptr, &tb, &rb) == -1);
	EXPECT(hoqp_solve_wbc_batched(nullptr, &wi, &wo, nullptr) == -1);
	EXPECT(hoqp_solve_wbc_unfused(nullptr, &wi, &wo, nullptr) == -1);
	EXPECT(hoqp_solve_wbc_noalloc(nullptr, &wi, &wo, nullptr) == -1);
	EXPECT(hoqp_solve_wbc_host(nullptr, &wi, &wo) == -1);
	EXPECT(hoqp_wbc_build_tasks_batched(nullptr, &wi, &tb, nullptr) == -1);
	EXPECT(hoqp_synchronize(nullptr, nullptr) == -1);
	EXPECT(hoqp_kernel_time_ms(nullptr, 0, nullptr, nullptr) == -1);
	EXPECT(hoqp_selftest(nullptr) == -1);
	// LDS sizing: valid extents, then every way of being out of range
	const int m[5] = {6, 0, 12, 18, 12}, p[5] = {0, 40, 0, 0, 0};
	EXPECT(hoqp_lds_bytes(30, 5, m, p) > 30000 && hoqp_lds_bytes(30, 5, m, p) < 65536);
	EXPECT(hoqp_lds_bytes(0, 5, m, p) == 0 && hoqp_lds_bytes(HOQP_MAX_VARS + 1, 5, m, p) == 0);
	EXPECT(hoqp_lds_bytes(30, 0, m, p) == 0 && hoqp_lds_bytes(30, HOQP_MAX_LEVELS + 1, m, p) == 0);
	EXPECT(hoqp_lds_bytes(30, 5, nullptr, p) == 0 && hoqp_lds_bytes(30, 5, m, nullptr) == 0);
	hoqp_quadruped_model qm;
	hoqp_default_quadruped_model(&qm);
	EXPECT(qm.base_mass > 0.0 && qm.gravity > 9.0);
	hoqp_gait g;
	EXPECT(hoqp_default_gait(0, &g) == 0 && g.n_steps > 0 && g.n_steps <= HOQP_MAX_GAIT_STEPS);
	EXPECT(hoqp_default_gait(1, &g) == 0 && hoqp_default_gait(7, &g) != 0);
	// sharded driver: argument checks
	hoqp_sharded *sh = nullptr;
	int dev[HOQP_MAX_SHARDS + 1] = {0};
	EXPECT(hoqp_sharded_create(nullptr, 1, dev, &sh) == -1);
	EXPECT(hoqp_sharded_create(&cfg, 0, dev, &sh) == -1);
	EXPECT(hoqp_sharded_create(&cfg, HOQP_MAX_SHARDS + 1, dev, &sh) == -1);
	EXPECT(hoqp_sharded_create(&cfg, 1, nullptr, &sh) == -1);
	EXPECT(hoqp_sharded_destroy(nullptr) == -1);
	EXPECT(hoqp_solve_wbc_sharded(nullptr, &wi, &wo, nullptr, nullptr) == -1);
	// with a device: the handle life cycle and the host-side checks behind it
	if (hoqp_create(&cfg, &h) == 0) {
		EXPECT(hoqp_solve_wbc_batched(h, nullptr, &wo, nullptr) == -1);
		EXPECT(hoqp_solve_wbc_batched(h, &wi, &wo, nullptr) == -1);  // null input arrays
		tb.n = HOQP_MAX_VARS + 1;
		EXPECT(hoqp_solve_tasks_batched(h, &tb, &rb, nullptr) == -1);
		tb.n = 8;
		tb.num_levels = HOQP_MAX_LEVELS + 1;
		EXPECT(hoqp_solve_tasks_batched(h, &tb, &rb, nullptr) == -1);
		EXPECT(std::strlen(hoqp_last_error_string(h)) > 0);
		EXPECT(hoqp_destroy(h) == 0);
		std::printf("abi_asan_check: handle life cycle exercised on device 0\n");
	}
	std::printf(bad ? "abi_asan_check: %d check(s) FAILED\n" : "abi_asan_check: ok\n", bad);
	return bad ? 1 : 0;
}
