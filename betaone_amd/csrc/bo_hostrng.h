// betaone_amd/csrc/bo_hostrng.h -- host-side, per-game random streams that are bit-compatible with
// numpy.random.RandomState(seed): MT19937 seeded by init_genrand, random_sample() = genrand_res53, and the
// LEGACY standard_gamma / dirichlet algorithms (numpy/random/src/legacy/legacy-distributions.c).
//
// What it replaces in the reference: the two NumPy RNG calls on the self-play path,
//   np.random.dirichlet([alpha]*n_legal)      /root/reference/mcts.py:192
//   np.random.choice(4672, p=...)             /root/reference/self_play.py:73 (via select_move_with_temperature :59-80)
// executed once per move and game.  With thousands of games per GPU a Python loop over RandomState objects
// costs 20-25 % of the step time; here the same streams are advanced natively (tests compare them with
// numpy draw for draw).  libm's pow/log are used where NumPy calls them.
#pragma once
#include <math.h>
#include <stdint.h>

struct HostRng {
    uint32_t key[624];
    int pos;
    int has_gauss;
    double gauss;
};

static inline void hr_seed(HostRng *s, uint32_t seed) {  // mt19937_seed == init_genrand
    for (int i = 0; i < 624; i++) {
        s->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    s->pos = 624;
    s->has_gauss = 0;
    s->gauss = 0.0;
}
static inline void hr_gen(HostRng *s) {
    const uint32_t N = 624, M = 397, MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
    uint32_t y;
    uint32_t i;
    for (i = 0; i < N - M; i++) {
        y = (s->key[i] & UPPER) | (s->key[i + 1] & LOWER);
        s->key[i] = s->key[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; i < N - 1; i++) {
        y = (s->key[i] & UPPER) | (s->key[i + 1] & LOWER);
        s->key[i] = s->key[i + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (s->key[N - 1] & UPPER) | (s->key[0] & LOWER);
    s->key[N - 1] = s->key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    s->pos = 0;
}
static inline uint32_t hr_u32(HostRng *s) {
    if (s->pos == 624) hr_gen(s);
    uint32_t y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
static inline double hr_double(HostRng *s) {  // RandomState.random_sample()
    int32_t a = (int32_t)(hr_u32(s) >> 5), b = (int32_t)(hr_u32(s) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}
static inline double hr_gauss(HostRng *s) {  // legacy_gauss
    if (s->has_gauss) {
        const double t = s->gauss;
        s->has_gauss = 0;
        s->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * hr_double(s) - 1.0;
        x2 = 2.0 * hr_double(s) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    s->gauss = f * x1;
    s->has_gauss = 1;
    return f * x2;
}
static inline double hr_std_exponential(HostRng *s) { return -log(1.0 - hr_double(s)); }
static inline double hr_std_gamma(HostRng *s, double shape) {  // legacy_standard_gamma
    double b, c, U, V, X, Y;
    if (shape == 1.0) return hr_std_exponential(s);
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
        for (;;) {
            U = hr_double(s);
            V = hr_std_exponential(s);
            if (U <= 1.0 - shape) {
                X = pow(U, 1. / shape);
                if (X <= V) return X;
            } else {
                Y = -log((1 - U) / shape);
                X = pow(1.0 - shape + shape * Y, 1. / shape);
                if (X <= (V + Y)) return X;
            }
        }
    }
    b = shape - 1. / 3.;
    c = 1. / sqrt(9 * b);
    for (;;) {
        do {
            X = hr_gauss(s);
            V = 1.0 + c * X;
        } while (V <= 0.0);
        V = V * V * V;
        U = hr_double(s);
        if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return (b * V);
        if (log(U) < 0.5 * X * X + b * (1. - V + log(V))) return (b * V);
    }
}
// RandomState.dirichlet([alpha]*n): out[0..n)
static inline void hr_dirichlet(HostRng *s, double alpha, int n, double *out) {
    double acc = 0.0;
    for (int j = 0; j < n; j++) {
        out[j] = hr_std_gamma(s, alpha);
        acc = acc + out[j];
    }
    const double invacc = 1 / acc;
    for (int j = 0; j < n; j++) out[j] = out[j] * invacc;
}

// select_move_with_temperature (self_play.py:59-80) for a pi with n <= 2 non-zero entries given as
// (action index, float32 probability) pairs; every operation on the zero entries of the dense vector is exact,
// so this equals the dense computation.  Returns the action index, or -1 when the caller must use the dense
// Python mirror (n == 0, n > 2, temperature 0, or a degenerate sum).
static inline int hr_select_action(HostRng *s, int n, const int32_t *idx, const float *val, int move_number, int threshold,
                                   double t_initial, double t_final) {
    const double temp = move_number < threshold ? t_initial : t_final;
    if (n < 1 || n > 2 || temp == 0.0) return -1;
    int i0 = idx[0], i1 = n == 2 ? idx[1] : -1;
    float p0 = val[0], p1 = n == 2 ? val[1] : 0.0f;
    if (n == 2 && i1 < i0) { int t = i0; i0 = i1; i1 = t; float q = p0; p0 = p1; p1 = q; }
    if (!(fabs(temp - 1.0) < 1e-6)) {  // apply_temperature, self_play.py:37-45
        double s0 = pow((double)p0, 1.0 / temp), s1 = n == 2 ? pow((double)p1, 1.0 / temp) : 0.0;
        if (!isfinite(s0)) s0 = 0.0;
        if (!isfinite(s1)) s1 = 0.0;
        const double sum = s0 + s1;
        if (!(sum > 1e-9)) return -1;
        p0 = (float)(s0 / sum);
        p1 = (float)(s1 / sum);
        const float rs = p0 + p1;
        if (fabsf(rs - 1.0f) > (float)1e-6 && rs > (float)1e-9) { p0 = p0 / rs; p1 = p1 / rs; }
    }
    const float prob_sum = p0 + p1;  // self_play.py:68-72
    if (fabsf(prob_sum - 1.0f) > (float)1e-6) {
        if (prob_sum > (float)1e-9) { p0 = p0 / prob_sum; p1 = p1 / prob_sum; }
        else return -1;
    }
    // RandomState.choice(4672, p=p): cdf = cumsum(double(p)); cdf /= cdf[-1]; searchsorted(random_sample(), 'right')
    const double d0 = (double)p0, d1 = (double)p1;
    if (fabs((d0 + d1) - 1.0) > 3.4e-4 || d0 < 0 || d1 < 0) return -1;
    const double c0 = d0, c1 = d0 + d1;
    const double last = n == 2 ? c1 : c0;
    const double u = hr_double(s);
    if (n == 1) return i0;
    return (c0 / last > u) ? i0 : i1;
}

// Same for any number of non-zero entries (FAST search mode: pi over all legal moves).  Entries are processed in
// action-index order; sums are sequential in double (not NumPy's pairwise order: the fast mode has no NumPy
// counterpart to be bit-compatible with).  One uniform draw, like RandomState.choice.
static inline int hr_select_action_general(HostRng *s, int n, const int32_t *idx, const float *val, int move_number,
                                           int threshold, double t_initial, double t_final) {
    if (n < 1 || n > 256) return -1;
    int order[256];
    for (int i = 0; i < n; i++) {
        int j = i;
        while (j > 0 && idx[order[j - 1]] > idx[i]) { order[j] = order[j - 1]; j--; }
        order[j] = i;
    }
    const double temp = move_number < threshold ? t_initial : t_final;
    double p[256], sum = 0.0;
    if (temp == 0.0) {
        int best = 0;
        for (int i = 1; i < n; i++) if (val[order[i]] > val[order[best]]) best = i;
        (void)hr_double(s);
        return idx[order[best]];
    }
    for (int i = 0; i < n; i++) {
        double x = (double)val[order[i]];
        if (!(fabs(temp - 1.0) < 1e-6)) { x = pow(x, 1.0 / temp); if (!isfinite(x)) x = 0.0; }
        p[i] = x;
        sum += x;
    }
    if (!(sum > 0.0)) return -1;
    const double u = hr_double(s);
    double c = 0.0;
    for (int i = 0; i < n; i++) {
        c += p[i] / sum;
        if (c > u) return idx[order[i]];
    }
    return idx[order[n - 1]];
}
