// phylo_train.h -- the host half of a VI training step in C++ (what phylo_amd/train.py does in NumPy, ~60 small array operations
// = 65 us of a 0.9 ms step): the reference's parameterisation (vcsmc.py:119-148), the chain rules from d logZ / d(lam, pi, Q) to
// its four variables, and the update rules of tf.train.GradientDescentOptimizer / AdamOptimizer (TF 1.15 defaults).  Plain host
// code, no HIP: the same formulas in the same order as train.py / model.py (tests/test_gpu_grad.py compares the two paths).
//   variables, packed: a_l[R] | a_r[R] | y_q[16] | y_station[4]        (R = N - 1; log-rates, vcsmc.py:119-124)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

// model.get_Q (vcsmc.py:138-148): off-diagonal = row softmax of y_q without the diagonal, diagonal = -row sum
inline void pt_get_Q(const double* y_q, double* Q) {
    for (int i = 0; i < 4; ++i) {
        double e[4], den = 0.0;
        for (int j = 0; j < 4; ++j) { e[j] = j == i ? 0.0 : std::exp(y_q[i * 4 + j]); den = den + e[j]; }
        const double inv = 1.0 / den;
        double rs = 0.0;
        for (int j = 0; j < 4; ++j) { Q[i * 4 + j] = e[j] * inv; rs = rs + Q[i * 4 + j]; }
        Q[i * 4 + i] = -rs;
    }
}
inline void pt_jc_Q(double* Q) {                          // vcsmc.py:126-129
    for (int i = 0; i < 16; ++i) Q[i] = 0.0 + 1.0 / 4.0;
    for (int i = 0; i < 4; ++i) Q[i * 4 + i] = -(4.0 - 1.0) / 4.0;
}
inline void pt_get_pi(const double* y_station, double* pi) {   // softmax (vcsmc.py:133-136)
    double e[4], s = 0.0;
    for (int j = 0; j < 4; ++j) { e[j] = std::exp(y_station[j]); s = s + e[j]; }
    for (int j = 0; j < 4; ++j) pi[j] = e[j] / s;
}
// train.chain_rules: grads packed like the variables; y_q / y_station get zeros under the JC69 model (they are constants there)
inline void pt_chain_rules(int R, int jc, const double* Q, const double* pi, const double* lam_l, const double* lam_r, const double* d_lam_l,
                           const double* d_lam_r, const double* d_pi, const double* d_Q, double* g) {
    for (int r = 0; r < R; ++r) { g[r] = d_lam_l[r] * lam_l[r]; g[R + r] = d_lam_r[r] * lam_r[r]; }
    double* gq = g + 2 * R;
    double* gs = gq + 16;
    memset(gq, 0, 20 * sizeof(double));
    if (jc) return;
    double dot = 0.0;
    for (int j = 0; j < 4; ++j) dot = dot + pi[j] * d_pi[j];
    for (int j = 0; j < 4; ++j) gs[j] = pi[j] * (d_pi[j] - dot);
    for (int i = 0; i < 4; ++i) {
        double q[4], dq[4], s = 0.0;
        for (int j = 0; j < 4; ++j) {
            q[j] = j == i ? 0.0 : Q[i * 4 + j];
            dq[j] = j == i ? 0.0 : d_Q[i * 4 + j] - d_Q[i * 4 + i];
            s = s + q[j] * dq[j];
        }
        for (int j = 0; j < 4; ++j) gq[i * 4 + j] = q[j] * (dq[j] - s);
    }
}
// optimiser state m, v: packed like the variables; only the first n entries are variables of the model (2R under JC69, 2R + 20 else)
inline void pt_apply(int n, double* vars, const double* grads_logZ, int kind, double lr, double b1, double b2, double eps, int64_t* t, double* m,
                     double* v) {
    if (kind == 0) {                                       // var <- var - lr d cost / d var, cost = -logZ
        for (int i = 0; i < n; ++i) vars[i] = vars[i] + lr * grads_logZ[i];
        return;
    }
    *t += 1;
    const double lr_t = lr * std::sqrt(1.0 - std::pow(b2, (double)*t)) / (1.0 - std::pow(b1, (double)*t));
    for (int i = 0; i < n; ++i) {
        const double g = -grads_logZ[i];
        m[i] = m[i] * b1 + (1.0 - b1) * g;
        v[i] = v[i] * b2 + (1.0 - b2) * g * g;
        vars[i] = vars[i] - lr_t * m[i] / (std::sqrt(v[i]) + eps);
    }
}
