/* CPU ORACLE (C) -- TEST INFRASTRUCTURE ONLY.  Never linked into or loaded by the product (pvsim / libpvsim_hip).
 *
 * Plain-C restatement of the VLAD + retrieval hot path of the reference, used (a) by tests as a second,
 * independent checker beside oracle/pvsim_oracle.py and (b) by bench.py's `cpu_baseline` leg, where it is the
 * thing timed on the GPU box's host cores (kind = "port").  Parity status: PINNED -- tests/test_oracle_golden.py
 * checks it against the golden vectors produced by running the reference (tests/golden/make_golden.py).
 *
 * Reference semantics followed (paths relative to the reference root; sklearn = scikit-learn 1.7.2):
 *   rootsift      pyvisim/features/_features.py:112-114
 *   assign        pyvisim/encoders/vlad.py:95 -> sklearn/cluster/_k_means_lloyd.pyx:168-218
 *                 (||c||^2 - 2 x.c in fp32, strict '<' scan: first minimum wins)
 *   aggregate     vlad.py:98-104 (sequential fp32 residual sums in descriptor order)
 *   normalise     vlad.py:106-111 (sign|v|^p, per-cluster norm + eps, divide)
 *   cosine        pyvisim/_utils.py:312-330 -> sklearn normalize + X @ Y^T
 *   top-k         pyvisim/eval.py:37-43 (argsort(-scores)[:k]; ties broken by index here)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EXPORT __attribute__((visibility("default")))

EXPORT int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* raw SIFT row (uint8) -> RootSIFT fp32 */
static void rootsift_row(const uint8_t* raw, int D, float* out) {
  float s = 0.f;
  for (int d = 0; d < D; ++d) s += (float)raw[d];
  const float den = s + 1e-7f;
  for (int d = 0; d < D; ++d) out[d] = sqrtf((float)raw[d] / den);
}

static int assign_one(const float* x, const float* C, const float* cn, int K, int D) {
  int best = 0;
  float bv = INFINITY;
  for (int k = 0; k < K; ++k) {
    const float* c = C + (size_t)k * D;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int d = 0;
    for (; d + 8 <= D; d += 8)
      for (int u = 0; u < 8; ++u) acc[u] += x[d + u] * c[d + u];
    float dot = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (; d < D; ++d) dot += x[d] * c[d];
    const float v = cn[k] + (-2.0f) * dot;
    if (v < bv) { bv = v; best = k; }
  }
  return best;
}

static float norm_ord(const float* v, int D, double ord) {
  if (ord == 2.0) { float s = 0.f; for (int d = 0; d < D; ++d) s += v[d] * v[d]; return sqrtf(s); }
  if (ord == 1.0) { float s = 0.f; for (int d = 0; d < D; ++d) s += fabsf(v[d]); return s; }
  if (isinf(ord)) { float m = 0.f; for (int d = 0; d < D; ++d) m = fmaxf(m, fabsf(v[d])); return m; }
  float s = 0.f;
  for (int d = 0; d < D; ++d) s += powf(fabsf(v[d]), (float)ord);
  return powf(s, (float)(1.0 / ord));
}

/* desc: packed rows [offsets[n_images]][D]; is_u8 != 0: uint8 raw SIFT with RootSIFT applied here.
 * out: [n_images][K*D]; labels (optional): [total].  threads <= 0: all available. */
EXPORT int orc_vlad_encode(const void* desc, int is_u8, const int64_t* offsets, int64_t n_images, const float* C,
                           int K, int D, double power, double ord, double eps, float* out, int32_t* labels,
                           int threads) {
  float* cn = (float*)malloc(sizeof(float) * (size_t)K);
  if (!cn) return 1;
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += C[(size_t)k * D + d] * C[(size_t)k * D + d];
    cn[k] = s;
  }
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
  {
    float* x = (float*)malloc(sizeof(float) * (size_t)D);
#pragma omp for schedule(dynamic, 4)
    for (int64_t img = 0; img < n_images; ++img) {
      float* V = out + (size_t)img * K * D;
      memset(V, 0, sizeof(float) * (size_t)K * D);
      for (int64_t r = offsets[img]; r < offsets[img + 1]; ++r) {
        const float* xr;
        if (is_u8) { rootsift_row((const uint8_t*)desc + (size_t)r * D, D, x); xr = x; }
        else xr = (const float*)desc + (size_t)r * D;
        const int l = assign_one(xr, C, cn, K, D);
        if (labels) labels[r] = l;
        float* v = V + (size_t)l * D;
        const float* c = C + (size_t)l * D;
        for (int d = 0; d < D; ++d) v[d] += (xr[d] - c[d]);
      }
      for (int k = 0; k < K; ++k) {
        float* v = V + (size_t)k * D;
        if (power != 1.0)
          for (int d = 0; d < D; ++d) {
            const float a = powf(fabsf(v[d]), (float)power);
            v[d] = v[d] > 0.f ? a : (v[d] < 0.f ? -a : 0.f * a);
          }
        const float den = norm_ord(v, D, ord) + (float)eps;
        for (int d = 0; d < D; ++d) v[d] = v[d] / den;
      }
    }
    free(x);
  }
  free(cn);
  return 0;
}

typedef struct { float s; int64_t i; } pair_t;
static int cmp_desc(const void* a, const void* b) {
  const pair_t *p = (const pair_t*)a, *q = (const pair_t*)b;
  const int pn = isnan(p->s), qn = isnan(q->s);
  if (pn != qn) return pn - qn;               /* NaN last */
  if (!pn) { if (p->s > q->s) return -1; if (p->s < q->s) return 1; }
  return (p->i > q->i) - (p->i < q->i);       /* ties by index */
}

/* per query: normalise the query and the WHOLE database (as sklearn does on every call, eval.py:75,131),
 * dot products, full sort, first k.  idx [nq][k], val [nq][k]. */
EXPORT int orc_retrieve(const float* Q, int64_t nq, const float* DB, int64_t N, int64_t L, int k, int64_t* idx,
                        float* val, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  int fail = 0;
#pragma omp parallel
  {
    float* dbn = (float*)malloc(sizeof(float) * (size_t)N * L);
    float* qn = (float*)malloc(sizeof(float) * (size_t)L);
    pair_t* pr = (pair_t*)malloc(sizeof(pair_t) * (size_t)N);
    if (!dbn || !qn || !pr) fail = 1;
#pragma omp for schedule(dynamic, 1)
    for (int64_t q = 0; q < nq; ++q) {
      if (fail) continue;
      const float* x = Q + (size_t)q * L;
      float s = 0.f;
      for (int64_t d = 0; d < L; ++d) s += x[d] * x[d];
      float nr = sqrtf(s); if (nr == 0.f) nr = 1.f;
      for (int64_t d = 0; d < L; ++d) qn[d] = x[d] / nr;
      for (int64_t j = 0; j < N; ++j) {
        const float* y = DB + (size_t)j * L;
        float* yn = dbn + (size_t)j * L;
        float t = 0.f;
        for (int64_t d = 0; d < L; ++d) t += y[d] * y[d];
        float ny = sqrtf(t); if (ny == 0.f) ny = 1.f;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t d = 0;
        for (; d + 8 <= L; d += 8)
          for (int u = 0; u < 8; ++u) { yn[d + u] = y[d + u] / ny; acc[u] += qn[d + u] * yn[d + u]; }
        float dot = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        for (; d < L; ++d) { yn[d] = y[d] / ny; dot += qn[d] * yn[d]; }
        pr[j].s = dot; pr[j].i = j;
      }
      qsort(pr, (size_t)N, sizeof(pair_t), cmp_desc);
      for (int r = 0; r < k; ++r) {
        idx[(size_t)q * k + r] = r < N ? pr[r].i : -1;
        val[(size_t)q * k + r] = r < N ? pr[r].s : -INFINITY;
      }
    }
    free(dbn); free(qn); free(pr);
  }
  return fail;
}

/* The device's cosine score as a DEFINED fp32 recurrence (not a reference function: the reference leaves the summation
 * order to BLAS).  The f32 MFMA adds its two products with two fused multiply-adds in lane-half order, so the exact GEMM
 * kernel computes, for one pair of rows,
 *     acc = fmaf(a[k], b[k], acc)   over k in the order (8t + e, 8t + 4 + e), e = 0..3, t = 0, 1, ...
 * in chains of 1024 k whose sums are added in order, then (0 + total) * (inv_a * inv_b)   (python-visual-similarity_amd/
 * csrc/gemm_mfma.hpp; checked on MI355X by csrc/bench/gemm_variants.hip "chain").  Lets the tests assert the device
 * cosine bit for bit instead of within a tolerance.  pairs: n x (i, j); out[n]. */
EXPORT int orc_cosine_chain(const float* A, const float* B, int64_t L, const float* inva, const float* invb,
                            const int64_t* pairs, int64_t n, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < n; ++p) {
    const float* a = A + (size_t)pairs[2 * p] * L;
    const float* b = B + (size_t)pairs[2 * p + 1] * L;
    float tot = 0.f;
    for (int64_t c0 = 0; c0 < L; c0 += 1024) {
      float acc = 0.f;
      const int64_t c1 = c0 + 1024 < L ? c0 + 1024 : L;
      for (int64_t b8 = c0; b8 < c1; b8 += 8)
        for (int e = 0; e < 4; ++e) {
          const int64_t k1 = b8 + e, k2 = b8 + 4 + e;
          if (k1 < L) acc = fmaf(a[k1], b[k1], acc);
          if (k2 < L) acc = fmaf(a[k2], b[k2], acc);
        }
      tot += acc;
    }
    const float sa = inva ? inva[pairs[2 * p]] : 1.f, sb = invb ? invb[pairs[2 * p + 1]] : 1.f;
    out[p] = (0.f + tot) * (sa * sb);
  }
  return 0;
}

/* K1 of the device as a defined recurrence: dot_j = fma chain over the dims in the exact kernel's MFMA order
 * ((8t + e, 8t + 4 + e), e = 0..3), v_j = fmaf(-2, dot_j, |c_j|^2) with |c_j|^2 the sequential fp32 sum the library
 * forms at table creation; strict '<' scan in ascending j.  labels[n]. */
EXPORT int orc_assign_chain(const float* X, int64_t n, const float* C, int K, int D, int32_t* labels) {
  float* cn = (float*)malloc(sizeof(float) * (size_t)K);
  if (!cn) return 1;
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += C[(size_t)k * D + d] * C[(size_t)k * D + d];
    cn[k] = s;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float* x = X + (size_t)i * D;
    float best = INFINITY;
    int bi = 0;
    for (int k = 0; k < K; ++k) {
      const float* c = C + (size_t)k * D;
      float acc = 0.f;
      for (int b8 = 0; b8 < D; b8 += 8)
        for (int e = 0; e < 4; ++e) {
          const int k1 = b8 + e, k2 = b8 + 4 + e;
          if (k1 < D) acc = fmaf(c[k1], x[k1], acc);
          if (k2 < D) acc = fmaf(c[k2], x[k2], acc);
        }
      const float v = fmaf(-2.f, acc, cn[k]);
      if (v < best) { best = v; bi = k; }
    }
    labels[i] = bi;
  }
  free(cn);
  return 0;
}
