"""CPU restatement of the base CoM spline planner -- TEST INFRASTRUCTURE ONLY.

Follows BasePlanner::FormulateOptimizationProblem (base_planner.cpp:162-172) literally and solves
the resulting equality-constrained QP by its KKT system; parity unpinned (the reference delegates
the solve to Drake, absent here; it ships no fixtures).

  * decision variables: per segment k a 2 x (degree+1) coefficient matrix C_k (InitDecisionVariables,
    :181-196); the program sees it through alpha_k = Map(C_k.data()), i.e. in COLUMN-MAJOR order
    (:226-228, :373-375), while the row selector T(t) = [v(t)' 0; 0 v(t)'] (GetTransformationMatrixAtT,
    :319-337) treats alpha_k as [x coefficients; y coefficients];
  * cost: sum over segments and over t = 0, 0.1, ... (while t < 1, accumulated in floating point:
    eleven samples) of dt * |T_ddot(t) alpha_k|^2 (AddAccelerationCost, :210-234);
  * constraints: position and velocity continuity at the 9 junctions (:236-250), initial position =
    current, final position = the given target (:252-266);
  * evaluation multiplies the coefficient MATRIX with the monomials (GeneratePolynomials, :77-98,
    EvalWalkTrajAtT, :100-142), i.e. reads the solution in row layout although it was optimised in
    the column-major layout: x(t) = sum_d C(0,d) t^d with C(0,d) = alpha[2d].  `fix_layout=True`
    reads it the way it was optimised.
"""
import numpy as np


def _basis(t, degree, deriv):
    v = np.zeros(degree + 1)
    for d in range(deriv, degree + 1):
        c = 1.0
        for j in range(deriv):
            c *= d - j
        v[d] = c * t ** (d - deriv)
    return v


def _T(t, degree, deriv):
    v = _basis(t, degree, deriv)
    T = np.zeros((2, 2 * (degree + 1)))
    T[0, :degree + 1] = v
    T[1, degree + 1:] = v
    return T


def base_walk_plan(curr_pos, final_pos, n_segments=10, degree=5):
    """alpha [n_segments, 2*(degree+1)]: the QP solution in the program's own variable order."""
    nv = 2 * (degree + 1)
    N = n_segments * nv
    H = np.zeros((N, N))
    dt, t = 0.1, 0.0
    Q = np.zeros((nv, nv))
    while t < 1:                      # :213-216, eleven samples in floating point
        Tdd = _T(t, degree, 2)
        Q += dt * Tdd.T @ Tdd
        t += dt
    for k in range(n_segments):
        H[k * nv:(k + 1) * nv, k * nv:(k + 1) * nv] = Q
    rows, rhs = [], []
    for k in range(n_segments - 1):
        for deriv in (0, 1):
            E = np.zeros((2, N))
            E[:, k * nv:(k + 1) * nv] = _T(1.0, degree, deriv)
            E[:, (k + 1) * nv:(k + 2) * nv] = -_T(0.0, degree, deriv)
            rows.append(E)
            rhs.append(np.zeros(2))
    E = np.zeros((2, N))
    E[:, :nv] = _T(0.0, degree, 0)
    rows.append(E)
    rhs.append(np.asarray(curr_pos, dtype=float))
    E = np.zeros((2, N))
    E[:, (n_segments - 1) * nv:] = _T(1.0, degree, 0)
    rows.append(E)
    rhs.append(np.asarray(final_pos, dtype=float))
    E, d = np.vstack(rows), np.concatenate(rhs)
    K = np.block([[2.0 * H, E.T], [E, np.zeros((E.shape[0], E.shape[0]))]])
    assert np.linalg.matrix_rank(K) == K.shape[0]          # the program has a unique solution
    sol = np.linalg.solve(K, np.concatenate([np.zeros(N), d]))
    return sol[:N].reshape(n_segments, nv)


def base_walk_eval(alpha, t, walking_height, fix_layout=False):
    """base_cmd (9,) = [pos; vel; acc] of EvalWalkTraj{Pos,Vel,Acc}AtT at time t."""
    n_segments, nv = alpha.shape
    degree = nv // 2 - 1
    k = 0
    while (k + 1 < t) and (k + 1 < n_segments):   # :103-106
        k += 1
    tin = t - k
    if fix_layout:
        C = alpha[k].reshape(2, degree + 1)                 # [x coefficients; y coefficients]
    else:
        C = alpha[k].reshape(degree + 1, 2).T               # Eigen column-major 2 x (degree+1)
    out = np.zeros(9)
    for deriv in range(3):
        out[3 * deriv:3 * deriv + 2] = C @ _basis(tin, degree, deriv)
    out[2] = walking_height
    return out
