// C-ABI of libpvsim_hip.so (declared in include/pvsim.h): context, tables, and the host-pointer /
// device-pointer entry points that sequence the kernels of vlad.hip, fisher.hip, cosine.hip, topk.hip.
#include <algorithm>
#include <cmath>
#include <limits>
#include <memory>

#include "common.hpp"

namespace pvs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int ws_reserve(pvs_ctx* ctx, int which, size_t bytes, void** out) {
  if (bytes == 0) bytes = 16;
  if (ctx->ws_bytes[which] < bytes) {
    if (ctx->ws[which]) {
      PVS_HIP(hipStreamSynchronize(ctx->stream));  // earlier work may still read the old block
      PVS_HIP(hipFree(ctx->ws[which]));
      ctx->ws[which] = nullptr;
      ctx->ws_bytes[which] = 0;
    }
    const size_t want = bytes + bytes / 8;
    PVS_HIP(hipMalloc(&ctx->ws[which], want));
    ctx->ws_bytes[which] = want;
  }
  *out = ctx->ws[which];
  return PVS_OK;
}

static int drain_timers(pvs_ctx* ctx) {
  if (ctx->pending.empty()) return PVS_OK;
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      ctx->t_total[r.slot] += ms;
      ctx->t_count[r.slot] += 1;
    }
    ctx->event_pool.push_back(r.a);
    ctx->event_pool.push_back(r.b);
  }
  ctx->pending.clear();
  return PVS_OK;
}

// ---- device blocks of table objects: recycled through the context (see pvs_ctx::block_cache)
static int table_alloc(pvs_ctx* ctx, void** dptr, size_t bytes) {
  const size_t want = (std::max<size_t>(bytes, 16) + 255) / 256 * 256;
  auto it = ctx->block_cache.lower_bound(want);
  if (it != ctx->block_cache.end() && it->first <= 2 * want + 4096) {
    *dptr = it->second;
    ctx->block_size[*dptr] = it->first;
    ctx->block_cache_bytes -= it->first;
    ctx->block_cache.erase(it);
    return PVS_OK;
  }
  PVS_HIP(hipMalloc(dptr, want));
  ctx->block_size[*dptr] = want;
  return PVS_OK;
}
static void table_free(pvs_ctx* ctx, void* p) {
  if (!p) return;
  if (ctx) {
    auto it = ctx->block_size.find(p);
    if (it != ctx->block_size.end()) {
      const size_t cap = it->second;
      ctx->block_size.erase(it);
      if (ctx->block_cache.size() < 96 && ctx->block_cache_bytes + cap <= ((size_t)1 << 30)) {
        ctx->block_cache.emplace(cap, p);
        ctx->block_cache_bytes += cap;
        return;
      }
    }
  }
  (void)hipFree(p);
}

template <typename T>
static int upload(pvs_ctx* ctx, T** dptr, const T* host, size_t count) {
  PVS_TRY(table_alloc(ctx, reinterpret_cast<void**>(dptr), std::max<size_t>(count, 1) * sizeof(T)));
  PVS_HIP(hipMemcpyAsync(*dptr, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return PVS_OK;
}

static size_t desc_elem_size(int kind) { return kind == PVS_DESC_U8_ROOTSIFT ? 1 : 4; }

static int check_kind(int kind) {
  if (kind != PVS_DESC_F32 && kind != PVS_DESC_F32_ROOTSIFT && kind != PVS_DESC_U8_ROOTSIFT)
    PVS_FAIL(PVS_ERR_INVALID, "unknown descriptor kind %d", kind);
  return PVS_OK;
}

int assign_tiles_for(int K);  // vlad.hip

}  // namespace pvs

using namespace pvs;

#define PVS_NEED(p, what) \
  if (!(p)) PVS_FAIL(PVS_ERR_INVALID, "%s: null %s", __func__, what)

// ================================================================================ context
PVS_EXPORT int pvs_version(void) { return PVS_VERSION; }
PVS_EXPORT const char* pvs_last_error(void) { return g_err; }

PVS_EXPORT int pvs_device_count(int* count) {
  PVS_NEED(count, "count");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return PVS_OK;
}

// Paths of the HIP runtime images mapped into this process (Linux: /proc/self/maps).  Two of them -- e.g. the system runtime
// this library was linked against plus the copy a PyTorch wheel bundles -- each open the device on their own and the second
// one finds no GPU; device pointers and streams cannot cross between them either.
static int count_hip_runtimes(char* first, size_t flen, char* second, size_t slen) {
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) return 0;
  char line[4096];
  int n = 0;
  first[0] = second[0] = 0;
  while (fgets(line, sizeof(line), f)) {
    char* p = strchr(line, '/');
    if (!p) continue;
    char* nl = strchr(p, '\n');
    if (nl) *nl = 0;
    const char* base = strrchr(p, '/');
    if (!base || !strstr(base, "libamdhip64")) continue;
    if (n >= 1 && strcmp(first, p) == 0) continue;
    if (n >= 2 && strcmp(second, p) == 0) continue;
    if (n == 0) snprintf(first, flen, "%s", p);
    else if (n == 1) snprintf(second, slen, "%s", p);
    ++n;
  }
  fclose(f);
  return n;
}

PVS_EXPORT int pvs_init(int device_id, void* stream, pvs_ctx** out) {
  PVS_NEED(out, "out");
  {
    char a[512], b[512];
    if (count_hip_runtimes(a, sizeof(a), b, sizeof(b)) > 1)
      PVS_FAIL(PVS_ERR_NO_DEVICE, "two HIP runtimes are mapped in this process (%s and %s): load libpvsim_hip.so after the runtime the "
               "rest of the process uses (the pvsim Python package does this by itself)", a, b);
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    PVS_FAIL(PVS_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU fallback");
  if (device_id < 0 || device_id >= n) PVS_FAIL(PVS_ERR_INVALID, "device %d out of range (0..%d)", device_id, n - 1);
  PVS_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  PVS_HIP(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    PVS_FAIL(PVS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id,
             prop.gcnArchName);
  std::unique_ptr<pvs_ctx> c(new pvs_ctx());
  c->device = device_id;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
  } else {
    PVS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->owns_stream = true;
  }
  *out = c.release();
  return PVS_OK;
}

PVS_EXPORT int pvs_destroy(pvs_ctx* ctx) {
  if (!ctx) return PVS_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  drain_timers(ctx);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  ctx->event_pool.clear();
  for (auto& kv : ctx->block_cache) (void)hipFree(kv.second);
  ctx->block_cache.clear();
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  for (int i = 0; i < pvs_ctx::NWS; ++i)
    if (ctx->ws[i]) hipFree(ctx->ws[i]);
  for (auto& p : ctx->gemm_plan)
    if (p.d_tiles) hipFree(p.d_tiles);
  if (ctx->d_queue) hipFree(ctx->d_queue);
  if (ctx->d_fused_stamps) hipFree(ctx->d_fused_stamps);
  if (ctx->owns_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return PVS_OK;
}

PVS_EXPORT int pvs_sync(pvs_ctx* ctx) {
  PVS_NEED(ctx, "ctx");
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}

PVS_EXPORT void* pvs_stream(pvs_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

PVS_EXPORT int pvs_device_name(pvs_ctx* ctx, char* buf, size_t buflen) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(buf, "buf");
  hipDeviceProp_t prop;
  PVS_HIP(hipGetDeviceProperties(&prop, ctx->device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return PVS_OK;
}

PVS_EXPORT int pvs_set_option(pvs_ctx* ctx, int option, int value) {
  PVS_NEED(ctx, "ctx");
  if (option < 0 || option >= PVS_OPT_COUNT_) PVS_FAIL(PVS_ERR_INVALID, "unknown option %d", option);
  const int hi = option == PVS_OPT_ASSIGN_PREFILTER ? 4 : (option == PVS_OPT_VLAD_PATH || option == PVS_OPT_TOPK_SELECT_ONLY) ? 3 : ((option == PVS_OPT_AGG_VARIANT || option == PVS_OPT_FISHER_SCALE) ? 2 : 1);
  if (value < 0 || value > hi) PVS_FAIL(PVS_ERR_INVALID, "option %d: value %d out of range 0..%d", option, value, hi);
  ctx->opt[option] = value;
  return PVS_OK;
}
PVS_EXPORT int pvs_get_option(pvs_ctx* ctx, int option, int* value) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(value, "value");
  if (option < 0 || option >= PVS_OPT_COUNT_) PVS_FAIL(PVS_ERR_INVALID, "unknown option %d", option);
  *value = ctx->opt[option];
  return PVS_OK;
}

PVS_EXPORT int pvs_malloc(pvs_ctx* ctx, size_t bytes, void** dptr) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(dptr, "dptr");
  PVS_HIP(hipSetDevice(ctx->device));
  PVS_HIP(hipMalloc(dptr, bytes ? bytes : 16));
  return PVS_OK;
}
PVS_EXPORT int pvs_free(pvs_ctx* ctx, void* dptr) {
  PVS_NEED(ctx, "ctx");
  if (dptr) {
    PVS_HIP(hipStreamSynchronize(ctx->stream));
    PVS_HIP(hipFree(dptr));
  }
  return PVS_OK;
}
PVS_EXPORT int pvs_memcpy_h2d(pvs_ctx* ctx, void* dst, const void* src, size_t bytes) {
  PVS_NEED(ctx, "ctx");
  PVS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}
PVS_EXPORT int pvs_memcpy_d2h(pvs_ctx* ctx, void* dst, const void* src, size_t bytes) {
  PVS_NEED(ctx, "ctx");
  PVS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}
PVS_EXPORT int pvs_memset(pvs_ctx* ctx, void* dst, int value, size_t bytes) {
  PVS_NEED(ctx, "ctx");
  PVS_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
  return PVS_OK;
}

namespace pvs {
__global__ void fill32_kernel(uint32_t* p, int64_t n, uint32_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill64_kernel(uint64_t* p, int64_t n, uint64_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
}  // namespace pvs

PVS_EXPORT int pvs_fill_dev(pvs_ctx* ctx, void* dst, int64_t n, int elem_bytes, uint64_t pattern) {
  PVS_NEED(ctx, "ctx");
  if (n <= 0) return PVS_OK;
  PVS_NEED(dst, "dst");
  if (elem_bytes != 4 && elem_bytes != 8) PVS_FAIL(PVS_ERR_INVALID, "fill: 4- or 8-byte elements");
  PVS_HIP(hipSetDevice(ctx->device));
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->num_cu * 8);
  if (elem_bytes == 4) hipLaunchKernelGGL(fill32_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<uint32_t*>(dst), n, (uint32_t)pattern);
  else hipLaunchKernelGGL(fill64_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<uint64_t*>(dst), n, pattern);
  PVS_HIP(hipGetLastError());
  return PVS_OK;
}

PVS_EXPORT int pvs_stream_wait(pvs_ctx* waiter, pvs_ctx* signal) {
  PVS_NEED(waiter, "waiter");
  PVS_NEED(signal, "signal");
  if (waiter->device != signal->device) PVS_FAIL(PVS_ERR_INVALID, "pvs_stream_wait: contexts on different devices");
  if (waiter->stream == signal->stream) return PVS_OK;
  PVS_HIP(hipSetDevice(waiter->device));
  hipEvent_t ev;
  PVS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  PVS_HIP(hipEventRecord(ev, signal->stream));
  PVS_HIP(hipStreamWaitEvent(waiter->stream, ev, 0));
  PVS_HIP(hipEventDestroy(ev));     // released by the runtime once the recorded work has completed
  return PVS_OK;
}

// ================================================================================ tables
PVS_EXPORT int pvs_codebook_create(pvs_ctx* ctx, const float* centroids, int K, int D, pvs_codebook** out) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(centroids, "centroids");
  PVS_NEED(out, "out");
  if (K < 1 || D < 1) PVS_FAIL(PVS_ERR_INVALID, "codebook: K and D must be positive (K=%d, D=%d)", K, D);
  PVS_HIP(hipSetDevice(ctx->device));
  pvs_codebook* cb = new pvs_codebook();
  cb->K = K;
  cb->D = D;
  const int nt = assign_tiles_for(K);
  cb->K_pad = (K + 32 * nt - 1) / (32 * nt) * (32 * nt);
  cb->D_pad = (D + 7) / 8 * 8;
  std::vector<float> pad((size_t)cb->K_pad * cb->D_pad, 0.f), cn(cb->K_pad, std::numeric_limits<float>::infinity());
  for (int k = 0; k < K; ++k) {
    // ||c||^2 in fp32 (sklearn row_norms(squared=True) on the fp32 centres)
    float s = 0.f;
    for (int d = 0; d < D; ++d) {
      const float v = centroids[(size_t)k * D + d];
      pad[(size_t)k * cb->D_pad + d] = v;
      const float sq = v * v;   // two statements: a product rounded to fp32, then the add (never contracted into an fma)
      s += sq;
    }
    cn[k] = s;
  }
  int st = upload(ctx, &cb->d_cent, centroids, (size_t)K * D);
  if (st == PVS_OK) st = upload(ctx, &cb->d_cpad, pad.data(), pad.size());
  if (st == PVS_OK) st = upload(ctx, &cb->d_cnorm, cn.data(), cn.size());
  // fp16 copy for the assignment prefilter (vlad.hip): whole table in LDS, every centre inside the fp16 range
  float amax = 0.f, cmax2 = 0.f;
  for (size_t i = 0; i < (size_t)K * D; ++i) amax = std::max(amax, std::fabs(centroids[i]));
  for (int k = 0; k < K; ++k) cmax2 = std::max(cmax2, cn[k]);
  if (st == PVS_OK && cb->K_pad <= 256 && D <= 128 && amax > 1e-30f && amax <= 1e30f && std::isfinite(cmax2)) {
    cb->D_pad16 = (D + 15) / 16 * 16;
    cb->cmax = std::sqrt(cmax2) * (1.f + 1e-6f);
    int e = 0;
    (void)std::frexp(amax, &e);                 // amax = m 2^e, m in [0.5, 1)
    cb->c16_shift = 13 - e;
    const float sc = std::ldexp(1.f, cb->c16_shift);
    const size_t tab = (size_t)cb->K_pad * cb->D_pad16;
    std::vector<_Float16> h16(2 * tab, (_Float16)0.f);
    for (int k = 0; k < K; ++k)
      for (int d = 0; d < D; ++d) {
        const float v = centroids[(size_t)k * D + d] * sc;        // exact (power of two)
        const _Float16 hi = (_Float16)v;
        // within a group of 16 dims the table is stored in the order the assignment kernel's lanes hold a descriptor:
        // half-wave h keeps dims 4h..4h+3 and 8+4h..8+4h+3 (two adjacent float4 loads per lane pair)
        const int g = d & 15, hh = (g >> 2) & 1, pos = 8 * hh + (g & 3) + 4 * (g >> 3);
        const size_t o = (size_t)k * cb->D_pad16 + (d & ~15) + pos;
        h16[o] = hi;
        h16[tab + o] = (_Float16)(v - (float)hi);   // v - hi is exact in fp32
      }
    if (table_alloc(ctx, &cb->d_c16, h16.size() * 2) != PVS_OK) st = PVS_ERR_OOM;
    else if (hipMemcpyAsync(cb->d_c16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;
    if (st == PVS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;   // h16 goes out of scope
    // -|c|^2/2 as three exact fp16 pieces per cluster (the prefilter adds it on the matrix pipe, vlad.hip / vlad_fused.hip)
    if (st == PVS_OK) {
      std::vector<_Float16> pk((size_t)cb->K_pad * 4, (_Float16)0.f);
      float amaxa = 0.f;
      for (int k = 0; k < K; ++k) amaxa = std::max(amaxa, std::fabs(std::ldexp(-0.5f * cn[k], cb->c16_shift)));
      int ea = 0;
      (void)std::frexp(amaxa, &ea);
      cb->cn_e1 = ea - 13;
      for (int k = 0; k < cb->K_pad; ++k) {
        if (k < K) {
          const float av = std::ldexp(-0.5f * cn[k], cb->c16_shift - cb->cn_e1);   // exact: powers of two
          const _Float16 p1 = (_Float16)av;
          const float r1 = av - (float)p1;
          const _Float16 p2 = (_Float16)r1;
          const float r2 = r1 - (float)p2;
          pk[(size_t)k * 4 + 0] = p1; pk[(size_t)k * 4 + 1] = p2; pk[(size_t)k * 4 + 2] = (_Float16)r2;
        } else {
          pk[(size_t)k * 4 + 0] = (_Float16)(-65504.f);   // below every real score of a row the prefilter takes
        }
      }
      if (!(std::isfinite(amaxa) && amaxa > 0.f)) {       // degenerate table: no prefilter at all
        table_free(ctx, cb->d_c16);
        cb->d_c16 = nullptr;
      } else if (table_alloc(ctx, &cb->d_cnk, pk.size() * 2) != PVS_OK) st = PVS_ERR_OOM;
      else if (hipMemcpyAsync(cb->d_cnk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;
      if (st == PVS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;
    }
    // the fp16 tables once more in natural dim order for the fused encode (vlad_fused.hip: K_pad == 256 and D == 128 only)
    if (st == PVS_OK && cb->d_c16 != nullptr && cb->K_pad == 256 && D == 128) {
      std::vector<_Float16> n16((size_t)2 * 256 * 128, (_Float16)0.f);
      for (int k = 0; k < K; ++k)
        for (int d = 0; d < 128; ++d) {
          const float v = centroids[(size_t)k * D + d] * sc;
          const _Float16 hi = (_Float16)v;
          n16[(size_t)k * 128 + d] = hi;
          n16[(size_t)256 * 128 + (size_t)k * 128 + d] = (_Float16)(v - (float)hi);
        }
      if (table_alloc(ctx, &cb->d_c16n, n16.size() * 2) != PVS_OK) st = PVS_ERR_OOM;
      else if (hipMemcpyAsync(cb->d_c16n, n16.data(), n16.size() * 2, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;
      if (st == PVS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;
    }
  }
  if (st == PVS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;
  if (st != PVS_OK) {
    pvs_codebook_destroy(ctx, cb);
    return st;
  }
  *out = cb;
  return PVS_OK;
}

PVS_EXPORT int pvs_codebook_destroy(pvs_ctx* ctx, pvs_codebook* cb) {
  if (!cb) return PVS_OK;
  if (ctx) hipStreamSynchronize(ctx->stream);
  table_free(ctx, cb->d_cent);
  table_free(ctx, cb->d_cpad);
  table_free(ctx, cb->d_cnorm);
  table_free(ctx, cb->d_c16);
  table_free(ctx, cb->d_c16n);
  table_free(ctx, cb->d_cnk);
  delete cb;
  return PVS_OK;
}

PVS_EXPORT int pvs_gmm_create(pvs_ctx* ctx, const double* weights, const double* means, const double* covariances,
                              int K, int D, pvs_gmm** out) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(weights, "weights");
  PVS_NEED(means, "means");
  PVS_NEED(covariances, "covariances");
  PVS_NEED(out, "out");
  if (K < 1 || D < 1) PVS_FAIL(PVS_ERR_INVALID, "gmm: K and D must be positive (K=%d, D=%d)", K, D);
  PVS_HIP(hipSetDevice(ctx->device));
  pvs_gmm* g = new pvs_gmm();
  g->K = K;
  g->D = D;
  // derived tables, fp64, following sklearn/mixture/_gaussian_mixture.py:413-450,495-512 ('diag'):
  //   prec_chol = 1/sqrt(cov); precisions = prec_chol^2; log_det = sum log(prec_chol)
  //   const_k = -0.5*(D*log(2 pi) + sum_d mu^2 prec) + log_det + log(w_k)
  std::vector<double> prec((size_t)K * D), mup((size_t)K * D), cst(K), inv_mu((size_t)K * D), inv_sg((size_t)K * D);
  // descriptors are fp32: sklearn casts log(2 pi) to X's dtype (_gaussian_mixture.py:512); it cancels in predict_proba
  // but not in the EM lower bound
  // and forms n_features * log(2 pi) in that dtype: a float32 product, rounded once (exact when D is a power of two)
  const float l2pf = (float)std::log(2.0 * M_PI);
  const volatile float dl2pf = (float)D * l2pf;
  const double d_log2pi = (double)dl2pf;
  for (int k = 0; k < K; ++k) {
    double s = 0.0, ld = 0.0;
    for (int d = 0; d < D; ++d) {
      const size_t i = (size_t)k * D + d;
      const double pc = 1.0 / std::sqrt(covariances[i]);
      const double p = pc * pc;
      prec[i] = p;
      mup[i] = means[i] * p;
      s += means[i] * means[i] * p;
      ld += std::log(pc);
      const double sw = std::sqrt(weights[k]);
      inv_mu[i] = 1.0 / (sw * std::sqrt(covariances[i]));
      inv_sg[i] = 1.0 / ((std::sqrt(2.0) * sw) * covariances[i]);
    }
    cst[k] = -0.5 * (d_log2pi + s) + ld + std::log(weights[k]);
  }
  int st = upload(ctx, &g->d_w, weights, (size_t)K);
  if (st == PVS_OK) st = upload(ctx, &g->d_mu, means, (size_t)K * D);
  if (st == PVS_OK) st = upload(ctx, &g->d_cov, covariances, (size_t)K * D);
  if (st == PVS_OK) st = upload(ctx, &g->d_prec, prec.data(), prec.size());
  if (st == PVS_OK) st = upload(ctx, &g->d_mup, mup.data(), mup.size());
  if (st == PVS_OK) st = upload(ctx, &g->d_const, cst.data(), cst.size());
  if (st == PVS_OK) st = upload(ctx, &g->d_inv_mu, inv_mu.data(), inv_mu.size());
  if (st == PVS_OK) st = upload(ctx, &g->d_inv_sg, inv_sg.data(), inv_sg.size());
  if (st == PVS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;
  if (st != PVS_OK) {
    pvs_gmm_destroy(ctx, g);
    return st;
  }
  *out = g;
  return PVS_OK;
}

PVS_EXPORT int pvs_gmm_destroy(pvs_ctx* ctx, pvs_gmm* g) {
  if (!g) return PVS_OK;
  if (ctx) hipStreamSynchronize(ctx->stream);
  for (double* p : {g->d_w, g->d_mu, g->d_cov, g->d_prec, g->d_mup, g->d_const, g->d_inv_mu, g->d_inv_sg})
    table_free(ctx, p);
  delete g;
  return PVS_OK;
}

PVS_EXPORT int pvs_pca_create(pvs_ctx* ctx, const float* components, const float* mean, int n_components, int d_in,
                              pvs_pca** out) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(components, "components");
  PVS_NEED(mean, "mean");
  PVS_NEED(out, "out");
  if (n_components < 1 || d_in < 1) PVS_FAIL(PVS_ERR_INVALID, "pca: bad shape (%d, %d)", n_components, d_in);
  PVS_HIP(hipSetDevice(ctx->device));
  pvs_pca* p = new pvs_pca();
  p->C = n_components;
  p->Din = d_in;
  // offset = mean @ components^T, fp32 (sklearn/decomposition/_base.py:116-166)
  std::vector<float> off(n_components);
  for (int c = 0; c < n_components; ++c) {
    float s = 0.f;
    for (int d = 0; d < d_in; ++d) s += mean[d] * components[(size_t)c * d_in + d];
    off[c] = s;
  }
  int st = upload(ctx, &p->d_comp, components, (size_t)n_components * d_in);
  if (st == PVS_OK) st = upload(ctx, &p->d_off, off.data(), off.size());
  if (st == PVS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = PVS_ERR_NO_DEVICE;
  if (st != PVS_OK) {
    pvs_pca_destroy(ctx, p);
    return st;
  }
  *out = p;
  return PVS_OK;
}

PVS_EXPORT int pvs_pca_destroy(pvs_ctx* ctx, pvs_pca* p) {
  if (!p) return PVS_OK;
  if (ctx) hipStreamSynchronize(ctx->stream);
  table_free(ctx, p->d_comp);
  table_free(ctx, p->d_off);
  delete p;
  return PVS_OK;
}

// ================================================================================ VLAD
static int check_norm(const pvs_norm_params* prm) {
  if (!prm) PVS_FAIL(PVS_ERR_INVALID, "null norm params");
  if (std::isnan(prm->norm_order) || prm->norm_order <= 0.0)
    PVS_FAIL(PVS_ERR_UNSUPPORTED, "norm_order must be > 0 or +inf (got %g)", prm->norm_order);
  return PVS_OK;
}

// optional PCA prologue: returns the descriptor view the encoder kernels should read
static int project_if_needed(pvs_ctx* ctx, const pvs_pca* pca, int model_dim, const void*& d_desc, int& kind,
                             int64_t total, int& ld) {
  if (!pca) {
    ld = model_dim;
    return PVS_OK;
  }
  if (pca->C != model_dim)
    PVS_FAIL(PVS_ERR_DIM, "PCA outputs %d components but the clustering model expects %d", pca->C, model_dim);
  float* proj = nullptr;
  PVS_TRY(ws_reserve(ctx, 3, (size_t)std::max<int64_t>(total, 1) * pca->C * sizeof(float),
                     reinterpret_cast<void**>(&proj)));
  PVS_TRY(launch_pca(ctx, pca, d_desc, kind, total, proj));
  d_desc = proj;
  kind = PVS_DESC_F32;
  ld = pca->C;
  return PVS_OK;
}

PVS_EXPORT int pvs_kmeans_predict_dev(pvs_ctx* ctx, const pvs_codebook* cb, const void* d_desc, int desc_kind,
                                      int64_t total_desc, int32_t* d_labels) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(cb, "codebook");
  PVS_TRY(check_kind(desc_kind));
  if (total_desc < 0) PVS_FAIL(PVS_ERR_INVALID, "negative descriptor count");
  if (total_desc == 0) return PVS_OK;
  PVS_NEED(d_desc, "descriptors");
  PVS_NEED(d_labels, "labels");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_assign(ctx, cb, d_desc, desc_kind, total_desc, cb->D, d_labels);
}

PVS_EXPORT int pvs_vlad_encode_dev(pvs_ctx* ctx, const pvs_codebook* cb, const pvs_pca* pca, const void* d_desc,
                                   int desc_kind, const int64_t* d_offsets, int64_t n_images, int64_t total_desc,
                                   const pvs_norm_params* prm, float* d_out, int32_t* d_labels, float* d_inv_norm) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(cb, "codebook");
  PVS_TRY(check_kind(desc_kind));
  PVS_TRY(check_norm(prm));
  if (n_images < 0 || total_desc < 0) PVS_FAIL(PVS_ERR_INVALID, "negative sizes");
  if (n_images == 0) return PVS_OK;
  PVS_NEED(d_offsets, "offsets");
  PVS_NEED(d_out, "out");
  if (total_desc > 0) PVS_NEED(d_desc, "descriptors");
  PVS_HIP(hipSetDevice(ctx->device));
  int kind = desc_kind, ld = 0;
  const void* x = d_desc;
  PVS_TRY(project_if_needed(ctx, pca, cb->D, x, kind, total_desc, ld));
  // PVS_OPT_VLAD_PATH = 3: the one-read fused kernel (vlad_fused.hip).  It is bit-identical to the two-kernel path and moves
  // half the bytes, but its phases are latency-bound today, so the default stays assign + aggregate.
  const int path = ctx->opt[PVS_OPT_VLAD_PATH];
  if (path == 3) {
    if (!(ctx->opt[PVS_OPT_ASSIGN_PREFILTER] != 0 && vlad_fused_eligible(cb, x, kind, ld, d_out)))
      PVS_FAIL(PVS_ERR_UNSUPPORTED, "fused VLAD encode needs D = 128, 128 < K <= 256, aligned rows and the fp16 prefilter tables");
    return launch_vlad_fused(ctx, cb, x, kind, ld, d_offsets, n_images, *prm, d_out, d_labels, d_inv_norm);
  }
  int32_t* labels = d_labels;
  if (!labels)
    PVS_TRY(ws_reserve(ctx, 1, (size_t)std::max<int64_t>(total_desc, 1) * sizeof(int32_t),
                       reinterpret_cast<void**>(&labels)));
  const float2* rowstat = nullptr;   // uint8 rows: per-row (sum + 1e-7, reciprocal) left by the prefilter for the aggregate pass
  PVS_TRY(launch_assign(ctx, cb, x, kind, total_desc, ld, labels, &rowstat));
  return launch_vlad_aggregate(ctx, cb, x, kind, ld, d_offsets, n_images, labels, *prm, d_out, d_inv_norm, false, rowstat, total_desc);
}

static int host_total(const int64_t* offsets, int64_t n_images, int64_t* total) {
  if (offsets[0] != 0) PVS_FAIL(PVS_ERR_INVALID, "offsets[0] must be 0");
  for (int64_t i = 0; i < n_images; ++i)
    if (offsets[i + 1] < offsets[i]) PVS_FAIL(PVS_ERR_INVALID, "offsets must be non-decreasing");
  *total = offsets[n_images];
  return PVS_OK;
}

PVS_EXPORT int pvs_vlad_encode(pvs_ctx* ctx, const pvs_codebook* cb, const pvs_pca* pca, const void* desc,
                               int desc_kind, const int64_t* offsets, int64_t n_images, const pvs_norm_params* prm,
                               float* out, int32_t* out_labels) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(cb, "codebook");
  PVS_TRY(check_kind(desc_kind));
  if (n_images < 0) PVS_FAIL(PVS_ERR_INVALID, "negative image count");
  if (n_images == 0) return PVS_OK;
  PVS_NEED(offsets, "offsets");
  PVS_NEED(out, "out");
  int64_t total = 0;
  PVS_TRY(host_total(offsets, n_images, &total));
  if (total > 0) PVS_NEED(desc, "descriptors");
  PVS_HIP(hipSetDevice(ctx->device));
  const int d_in = pca ? pca->Din : cb->D;
  const size_t desc_bytes = (size_t)total * d_in * desc_elem_size(desc_kind);
  const size_t out_elems = (size_t)n_images * cb->K * cb->D;
  char* d_x = nullptr;
  char* d_small = nullptr;
  float* d_out = nullptr;
  PVS_TRY(ws_reserve(ctx, 0, desc_bytes, reinterpret_cast<void**>(&d_x)));
  const size_t off_bytes = (size_t)(n_images + 1) * sizeof(int64_t);
  const size_t lab_off = (off_bytes + 255) / 256 * 256;
  PVS_TRY(ws_reserve(ctx, 2, lab_off + (size_t)std::max<int64_t>(total, 1) * sizeof(int32_t) + out_elems * 4 + 256,
                     reinterpret_cast<void**>(&d_small)));
  int64_t* d_off = reinterpret_cast<int64_t*>(d_small);
  int32_t* d_lab = reinterpret_cast<int32_t*>(d_small + lab_off);
  const size_t out_off = (lab_off + (size_t)std::max<int64_t>(total, 1) * 4 + 255) / 256 * 256;
  d_out = reinterpret_cast<float*>(d_small + out_off);
  if (desc_bytes) PVS_HIP(hipMemcpyAsync(d_x, desc, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
  PVS_HIP(hipMemcpyAsync(d_off, offsets, off_bytes, hipMemcpyHostToDevice, ctx->stream));
  PVS_TRY(pvs_vlad_encode_dev(ctx, cb, pca, d_x, desc_kind, d_off, n_images, total, prm, d_out, d_lab, nullptr));
  PVS_HIP(hipMemcpyAsync(out, d_out, out_elems * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (out_labels && total > 0)
    PVS_HIP(hipMemcpyAsync(out_labels, d_lab, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}

// ================================================================================ Fisher / PCA
PVS_EXPORT int pvs_pca_transform_dev(pvs_ctx* ctx, const pvs_pca* p, const void* d_desc, int desc_kind,
                                     int64_t total_desc, float* d_out) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(p, "pca");
  PVS_TRY(check_kind(desc_kind));
  if (total_desc <= 0) return PVS_OK;
  PVS_NEED(d_desc, "descriptors");
  PVS_NEED(d_out, "out");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_pca(ctx, p, d_desc, desc_kind, total_desc, d_out);
}

PVS_EXPORT int pvs_gmm_predict_proba_dev(pvs_ctx* ctx, const pvs_gmm* g, const void* d_desc, int desc_kind,
                                         int64_t total_desc, double* d_resp) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(g, "gmm");
  PVS_TRY(check_kind(desc_kind));
  if (total_desc <= 0) return PVS_OK;
  PVS_NEED(d_desc, "descriptors");
  PVS_NEED(d_resp, "resp");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_gmm_posterior(ctx, g, d_desc, desc_kind, g->D, total_desc, d_resp);
}

PVS_EXPORT int pvs_fisher_encode_dev(pvs_ctx* ctx, const pvs_gmm* g, const pvs_pca* pca, const void* d_desc,
                                     int desc_kind, const int64_t* d_offsets, int64_t n_images, int64_t total_desc,
                                     const pvs_norm_params* prm, void* d_out, int out_f64) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(g, "gmm");
  PVS_TRY(check_kind(desc_kind));
  PVS_TRY(check_norm(prm));
  if (n_images < 0 || total_desc < 0) PVS_FAIL(PVS_ERR_INVALID, "negative sizes");
  if (n_images == 0) return PVS_OK;
  PVS_NEED(d_offsets, "offsets");
  PVS_NEED(d_out, "out");
  if (total_desc > 0) PVS_NEED(d_desc, "descriptors");
  PVS_HIP(hipSetDevice(ctx->device));
  int kind = desc_kind, ld = 0;
  const void* x = d_desc;
  PVS_TRY(project_if_needed(ctx, pca, g->D, x, kind, total_desc, ld));
  return launch_fisher(ctx, g, x, kind, ld, d_offsets, n_images, total_desc, *prm, d_out, out_f64);
}

PVS_EXPORT int pvs_fisher_encode(pvs_ctx* ctx, const pvs_gmm* g, const pvs_pca* pca, const void* desc, int desc_kind,
                                 const int64_t* offsets, int64_t n_images, const pvs_norm_params* prm, double* out) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(g, "gmm");
  PVS_TRY(check_kind(desc_kind));
  if (n_images < 0) PVS_FAIL(PVS_ERR_INVALID, "negative image count");
  if (n_images == 0) return PVS_OK;
  PVS_NEED(offsets, "offsets");
  PVS_NEED(out, "out");
  int64_t total = 0;
  PVS_TRY(host_total(offsets, n_images, &total));
  if (total > 0) PVS_NEED(desc, "descriptors");
  PVS_HIP(hipSetDevice(ctx->device));
  const int d_in = pca ? pca->Din : g->D;
  const size_t desc_bytes = (size_t)total * d_in * desc_elem_size(desc_kind);
  const size_t out_elems = (size_t)n_images * ((size_t)g->K + 2 * (size_t)g->K * g->D);
  char* d_x = nullptr;
  char* d_small = nullptr;
  PVS_TRY(ws_reserve(ctx, 0, desc_bytes, reinterpret_cast<void**>(&d_x)));
  const size_t off_bytes = ((size_t)(n_images + 1) * sizeof(int64_t) + 255) / 256 * 256;
  PVS_TRY(ws_reserve(ctx, 2, off_bytes + out_elems * 8, reinterpret_cast<void**>(&d_small)));
  int64_t* d_off = reinterpret_cast<int64_t*>(d_small);
  double* d_out = reinterpret_cast<double*>(d_small + off_bytes);
  if (desc_bytes) PVS_HIP(hipMemcpyAsync(d_x, desc, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
  PVS_HIP(hipMemcpyAsync(d_off, offsets, (size_t)(n_images + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  PVS_TRY(pvs_fisher_encode_dev(ctx, g, pca, d_x, desc_kind, d_off, n_images, total, prm, d_out, 1));
  PVS_HIP(hipMemcpyAsync(out, d_out, out_elems * 8, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}

// ================================================================================ cosine / top-k
PVS_EXPORT int pvs_row_inv_norms_dev(pvs_ctx* ctx, const float* d_x, int64_t rows, int64_t L, float* d_inv) {
  PVS_NEED(ctx, "ctx");
  if (rows <= 0) return PVS_OK;
  PVS_NEED(d_x, "x");
  PVS_NEED(d_inv, "inv");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_row_inv_norms(ctx, d_x, rows, L, d_inv);
}

PVS_EXPORT int pvs_cosine_dev(pvs_ctx* ctx, const float* d_A, int64_t M, const float* d_B, int64_t N, int64_t L,
                              const float* d_inv_a, const float* d_inv_b, float* d_out, int64_t ldo) {
  PVS_NEED(ctx, "ctx");
  if (M <= 0 || N <= 0) return PVS_OK;
  PVS_NEED(d_A, "A");
  PVS_NEED(d_B, "B");
  PVS_NEED(d_out, "out");
  if (ldo < N) PVS_FAIL(PVS_ERR_INVALID, "cosine: ldo (%lld) < N (%lld)", (long long)ldo, (long long)N);
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_cosine_f32(ctx, d_A, M, d_B, N, L, d_inv_a, d_inv_b, d_out, ldo);
}

PVS_EXPORT int pvs_cosine_dual_dev(pvs_ctx* ctx, const float* d_A, int64_t M, const float* d_B, int64_t N, int64_t L,
                                   const float* d_inv_a, const float* d_inv_b, float* d_out, int64_t ldo, float* d_out_t,
                                   int64_t ldt) {
  PVS_NEED(ctx, "ctx");
  if (M <= 0 || N <= 0) return PVS_OK;
  PVS_NEED(d_A, "A");
  PVS_NEED(d_B, "B");
  PVS_NEED(d_out, "out");
  PVS_NEED(d_out_t, "out_t");
  if (ldo < N || ldt < M) PVS_FAIL(PVS_ERR_INVALID, "cosine dual: ldo < N or ldt < M");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_cosine_f32_dual(ctx, d_A, M, d_B, N, L, d_inv_a, d_inv_b, d_out, ldo, d_out_t, ldt);
}

PVS_EXPORT int pvs_cosine(pvs_ctx* ctx, const void* A, int64_t M, const void* B, int64_t N, int64_t L, int is_f64,
                          void* out) {
  PVS_NEED(ctx, "ctx");
  if (M < 0 || N < 0) PVS_FAIL(PVS_ERR_INVALID, "negative sizes");
  if (L <= 1) PVS_FAIL(PVS_ERR_INVALID, "Cosine similarity requires at least 2 features. Got %lld features.", (long long)L);
  if (M == 0 || N == 0) return PVS_OK;
  PVS_NEED(A, "A");
  PVS_NEED(B, "B");
  PVS_NEED(out, "out");
  PVS_HIP(hipSetDevice(ctx->device));
  const size_t es = is_f64 ? 8 : 4;
  const size_t a_bytes = ((size_t)M * L * es + 255) / 256 * 256, b_bytes = ((size_t)N * L * es + 255) / 256 * 256;
  char* d_in = nullptr;
  char* d_o = nullptr;
  const bool same = (A == B && M == N);
  PVS_TRY(ws_reserve(ctx, 0, a_bytes + (same ? 0 : b_bytes), reinterpret_cast<void**>(&d_in)));
  const size_t nrm_bytes = ((size_t)(M + N) * 4 + 255) / 256 * 256;
  PVS_TRY(ws_reserve(ctx, 2, nrm_bytes + (size_t)M * N * es, reinterpret_cast<void**>(&d_o)));
  PVS_HIP(hipMemcpyAsync(d_in, A, (size_t)M * L * es, hipMemcpyHostToDevice, ctx->stream));
  char* d_b = d_in;
  if (!same) {
    d_b = d_in + a_bytes;
    PVS_HIP(hipMemcpyAsync(d_b, B, (size_t)N * L * es, hipMemcpyHostToDevice, ctx->stream));
  }
  void* d_out = d_o + nrm_bytes;
  if (is_f64) {
    PVS_TRY(launch_cosine_f64(ctx, reinterpret_cast<double*>(d_in), M, reinterpret_cast<double*>(d_b), N, L,
                              reinterpret_cast<double*>(d_out)));
  } else {
    float* inva = reinterpret_cast<float*>(d_o);
    float* invb = inva + M;
    PVS_TRY(launch_row_inv_norms(ctx, reinterpret_cast<float*>(d_in), M, L, inva));
    PVS_TRY(launch_row_inv_norms(ctx, reinterpret_cast<float*>(d_b), N, L, invb));
    PVS_TRY(launch_cosine_f32(ctx, reinterpret_cast<float*>(d_in), M, reinterpret_cast<float*>(d_b), N, L, inva, invb,
                              reinterpret_cast<float*>(d_out), N));
  }
  PVS_HIP(hipMemcpyAsync(out, d_out, (size_t)M * N * es, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}

PVS_EXPORT int pvs_topk_dev(pvs_ctx* ctx, const float* d_scores, int64_t nq, int64_t ncols, int64_t ld, int k,
                            int64_t col_offset, int merge, int64_t* d_idx, float* d_val) {
  PVS_NEED(ctx, "ctx");
  if (nq <= 0) return PVS_OK;
  PVS_NEED(d_idx, "idx");
  PVS_NEED(d_val, "val");
  if (ncols > 0) PVS_NEED(d_scores, "scores");
  if (ncols < 0 || ld < ncols || col_offset < 0) PVS_FAIL(PVS_ERR_INVALID, "top-k: bad panel geometry");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_topk(ctx, d_scores, nq, ncols, ld, k, col_offset, merge, d_idx, d_val);
}

PVS_EXPORT int pvs_topk_merge_dev(pvs_ctx* ctx, const int64_t* d_idx_lists, const float* d_val_lists, int n_lists,
                                  int64_t nq, int k, int64_t* d_idx, float* d_val) {
  PVS_NEED(ctx, "ctx");
  if (nq <= 0) return PVS_OK;
  PVS_NEED(d_idx_lists, "idx lists");
  PVS_NEED(d_val_lists, "val lists");
  PVS_NEED(d_idx, "idx");
  PVS_NEED(d_val, "val");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_topk_merge(ctx, d_idx_lists, d_val_lists, n_lists, nq, k, d_idx, d_val);
}

// GEMM panel + select, tiled over queries and database columns; `f16` selects the fp16-operand GEMM
static int cosine_topk_impl(pvs_ctx* ctx, const void* d_Q, int64_t nq, const void* d_DB, int64_t N, int64_t L, bool f16,
                            const float* d_inv_q, const float* d_inv_db, int k, int64_t col_offset, int merge,
                            int64_t* d_idx, float* d_val) {
  PVS_NEED(ctx, "ctx");
  if (nq <= 0) return PVS_OK;
  PVS_NEED(d_Q, "Q");
  PVS_NEED(d_idx, "idx");
  PVS_NEED(d_val, "val");
  if (N < 0) PVS_FAIL(PVS_ERR_INVALID, "negative N");
  PVS_HIP(hipSetDevice(ctx->device));
  if (N == 0) return launch_topk(ctx, nullptr, nq, 0, 0, k, col_offset, merge, d_idx, d_val);
  PVS_NEED(d_DB, "DB");
  // score panel: at most 8192 x 32768 fp32 = 1 GiB, written by the GEMM and consumed by the select.
  // Ranking deeper than 1024 pages through complete rows, so the panel then spans all N columns.
  const bool deep = k > 1024;
  if (deep && (merge || N > (int64_t)1 << 28)) PVS_FAIL(PVS_ERR_UNSUPPORTED, "ranking depth %d needs a single panel", k);
  // one or two queries (eval.retrieve_top_k_similar against a resident index): a 1 x N score row in ONE pass over the database at
  // HBM speed instead of 128-row MFMA tiles with 127 idle rows; the same scores bit for bit (filter.hip)
  if (!f16 && N <= ((int64_t)1 << 28) && cosine_dense_rows_eligible(static_cast<const float*>(d_Q), static_cast<const float*>(d_DB), nq, L)) {
    float* row = nullptr;
    PVS_TRY(ws_reserve(ctx, 2, (size_t)nq * N * sizeof(float), reinterpret_cast<void**>(&row)));
    PVS_TRY(launch_cosine_dense_rows(ctx, static_cast<const float*>(d_Q), nq, static_cast<const float*>(d_DB), N, L, d_inv_q, d_inv_db, row, N));
    return launch_topk(ctx, row, nq, N, N, k, col_offset, merge, d_idx, d_val);
  }
  const int64_t NC = deep ? N : std::min<int64_t>(N, 32768);
  const int64_t QT = deep ? std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)1 << 28) / N))
                          : std::min<int64_t>(nq, 8192);
  const size_t esz = f16 ? 2 : 4;
  const char* q = static_cast<const char*>(d_Q);
  const char* db = static_cast<const char*>(d_DB);
  float* panel = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, (size_t)QT * NC * sizeof(float), reinterpret_cast<void**>(&panel)));
  for (int64_t q0 = 0; q0 < nq; q0 += QT) {
    const int64_t qn = std::min(QT, nq - q0);
    for (int64_t c0 = 0; c0 < N; c0 += NC) {
      const int64_t cn = std::min(NC, N - c0);
      const float* iq = d_inv_q ? d_inv_q + q0 : nullptr;
      const float* idb = d_inv_db ? d_inv_db + c0 : nullptr;
      if (f16)
        PVS_TRY(launch_cosine_f16(ctx, q + (size_t)q0 * L * esz, qn, db + (size_t)c0 * L * esz, cn, L, iq, idb, panel, cn));
      else
        PVS_TRY(launch_cosine_f32(ctx, reinterpret_cast<const float*>(q) + q0 * L, qn,
                                  reinterpret_cast<const float*>(db) + c0 * L, cn, L, iq, idb, panel, cn));
      PVS_TRY(launch_topk(ctx, panel, qn, cn, cn, k, col_offset + c0, (merge || c0 > 0) ? 1 : 0, d_idx + q0 * k,
                          d_val + q0 * k));
    }
  }
  return PVS_OK;
}

PVS_EXPORT int pvs_cosine_topk_dev(pvs_ctx* ctx, const float* d_Q, int64_t nq, const float* d_DB, int64_t N, int64_t L,
                                   const float* d_inv_q, const float* d_inv_db, int k, int64_t col_offset, int merge,
                                   int64_t* d_idx, float* d_val) {
  return cosine_topk_impl(ctx, d_Q, nq, d_DB, N, L, false, d_inv_q, d_inv_db, k, col_offset, merge, d_idx, d_val);
}

namespace pvs {
int cosine_topk_exact(pvs_ctx* ctx, const float* d_Q, int64_t nq, const float* d_DB, int64_t N, int64_t L, const float* d_inv_q,
                      const float* d_inv_db, int k, int64_t* d_idx, float* d_val) {
  return cosine_topk_impl(ctx, d_Q, nq, d_DB, N, L, false, d_inv_q, d_inv_db, k, 0, 0, d_idx, d_val);
}
}  // namespace pvs

// Exact top-k through the fp16 prefilter + exact re-scoring (filter.hip); inputs that do not qualify take the plain path.
PVS_EXPORT int pvs_cosine_topk_filtered_dev(pvs_ctx* ctx, const float* d_Q, int64_t nq, const float* d_DB, int64_t N, int64_t L,
                                            const float* d_inv_q, const float* d_inv_db, int k, int64_t* d_idx, float* d_val,
                                            int64_t* h_stats) {
  PVS_NEED(ctx, "ctx");
  if (h_stats) h_stats[0] = h_stats[1] = h_stats[2] = h_stats[3] = 0;
  if (nq <= 0) return PVS_OK;
  PVS_NEED(d_Q, "Q");
  PVS_NEED(d_idx, "idx");
  PVS_NEED(d_val, "val");
  PVS_HIP(hipSetDevice(ctx->device));
  if (N > 0 && d_DB != nullptr) {
    const int rc = launch_cosine_topk_filtered(ctx, d_Q, nq, d_DB, N, L, d_inv_q, d_inv_db, k, 0, d_idx, d_val, h_stats);
    if (rc != PVS_ERR_UNSUPPORTED) return rc;
    if (h_stats) h_stats[0] = h_stats[1] = h_stats[2] = h_stats[3] = 0;
  }
  return cosine_topk_impl(ctx, d_Q, nq, d_DB, N, L, false, d_inv_q, d_inv_db, k, 0, 0, d_idx, d_val);
}

PVS_EXPORT int pvs_cosine_topk_f16_dev(pvs_ctx* ctx, const void* d_Q16, int64_t nq, const void* d_DB16, int64_t N,
                                       int64_t L, const float* d_inv_q, const float* d_inv_db, int k, int64_t col_offset,
                                       int merge, int64_t* d_idx, float* d_val) {
  return cosine_topk_impl(ctx, d_Q16, nq, d_DB16, N, L, true, d_inv_q, d_inv_db, k, col_offset, merge, d_idx, d_val);
}

PVS_EXPORT int pvs_f32_to_f16_dev(pvs_ctx* ctx, const float* d_src, int64_t n, void* d_dst) {
  PVS_NEED(ctx, "ctx");
  if (n <= 0) return PVS_OK;
  PVS_NEED(d_src, "src");
  PVS_NEED(d_dst, "dst");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_f32_to_f16(ctx, d_src, n, d_dst);
}

PVS_EXPORT int pvs_cosine_f16_dev(pvs_ctx* ctx, const void* d_A16, int64_t M, const void* d_B16, int64_t N, int64_t L,
                                  const float* d_inv_a, const float* d_inv_b, float* d_out, int64_t ldo) {
  PVS_NEED(ctx, "ctx");
  if (M <= 0 || N <= 0) return PVS_OK;
  PVS_NEED(d_A16, "A");
  PVS_NEED(d_B16, "B");
  PVS_NEED(d_out, "out");
  if (ldo < N) PVS_FAIL(PVS_ERR_INVALID, "cosine: ldo (%lld) < N (%lld)", (long long)ldo, (long long)N);
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_cosine_f16(ctx, d_A16, M, d_B16, N, L, d_inv_a, d_inv_b, d_out, ldo);
}

PVS_EXPORT int pvs_cosine_topk(pvs_ctx* ctx, const float* Q, int64_t nq, const float* DB, int64_t N, int64_t L, int k,
                               int64_t* out_idx, float* out_val) {
  PVS_NEED(ctx, "ctx");
  if (nq <= 0) return PVS_OK;
  if (L <= 1) PVS_FAIL(PVS_ERR_INVALID, "Cosine similarity requires at least 2 features. Got %lld features.", (long long)L);
  PVS_NEED(Q, "Q");
  PVS_NEED(DB, "DB");
  PVS_NEED(out_idx, "idx");
  PVS_NEED(out_val, "val");
  if (N <= 0) PVS_FAIL(PVS_ERR_INVALID, "empty database");
  PVS_HIP(hipSetDevice(ctx->device));
  const bool same = (Q == DB && nq == N);
  const size_t q_bytes = ((size_t)nq * L * 4 + 255) / 256 * 256, db_bytes = ((size_t)N * L * 4 + 255) / 256 * 256;
  char* d_in = nullptr;
  char* d_s = nullptr;
  PVS_TRY(ws_reserve(ctx, 0, q_bytes + (same ? 0 : db_bytes), reinterpret_cast<void**>(&d_in)));
  const size_t nrm = ((size_t)(nq + N) * 4 + 255) / 256 * 256, idxb = ((size_t)nq * k * 8 + 255) / 256 * 256;
  PVS_TRY(ws_reserve(ctx, 1, nrm + idxb + (size_t)nq * k * 4, reinterpret_cast<void**>(&d_s)));
  float* dq = reinterpret_cast<float*>(d_in);
  float* ddb = same ? dq : reinterpret_cast<float*>(d_in + q_bytes);
  PVS_HIP(hipMemcpyAsync(dq, Q, (size_t)nq * L * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!same) PVS_HIP(hipMemcpyAsync(ddb, DB, (size_t)N * L * 4, hipMemcpyHostToDevice, ctx->stream));
  float* invq = reinterpret_cast<float*>(d_s);
  float* invd = invq + nq;
  int64_t* d_idx = reinterpret_cast<int64_t*>(d_s + nrm);
  float* d_val = reinterpret_cast<float*>(d_s + nrm + idxb);
  PVS_TRY(launch_row_inv_norms(ctx, dq, nq, L, invq));
  PVS_TRY(launch_row_inv_norms(ctx, ddb, N, L, invd));
  // many queries: the filtered path returns the same lists (bit-identical) faster; it declines what does not qualify
  const float* invd_use = same ? invq : invd;   // one pointer for both operands of a self-similarity: the symmetric kernel applies
  if (nq >= 512) PVS_TRY(pvs_cosine_topk_filtered_dev(ctx, dq, nq, ddb, N, L, invq, invd_use, k, d_idx, d_val, nullptr));
  else PVS_TRY(pvs_cosine_topk_dev(ctx, dq, nq, ddb, N, L, invq, invd_use, k, 0, 0, d_idx, d_val));
  PVS_HIP(hipMemcpyAsync(out_idx, d_idx, (size_t)nq * k * 8, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipMemcpyAsync(out_val, d_val, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}


// ---------------------------------------------------------------- float64 scores + ranking (Fisher encodings)
PVS_EXPORT int pvs_row_inv_norms_f64_dev(pvs_ctx* ctx, const double* d_x, int64_t rows, int64_t L, double* d_inv) {
  PVS_NEED(ctx, "ctx");
  if (rows <= 0) return PVS_OK;
  PVS_NEED(d_x, "x");
  PVS_NEED(d_inv, "inv");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_row_inv_norms_f64(ctx, d_x, rows, L, d_inv);
}

PVS_EXPORT int pvs_cosine_f64_dev(pvs_ctx* ctx, const double* d_A, int64_t M, const double* d_B, int64_t N, int64_t L,
                                  const double* d_inv_a, const double* d_inv_b, double* d_out, int64_t ldo) {
  PVS_NEED(ctx, "ctx");
  if (M <= 0 || N <= 0) return PVS_OK;
  PVS_NEED(d_A, "A");
  PVS_NEED(d_B, "B");
  PVS_NEED(d_out, "out");
  if (ldo < N) PVS_FAIL(PVS_ERR_INVALID, "cosine: ldo (%lld) < N (%lld)", (long long)ldo, (long long)N);
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_cosine_f64_dev(ctx, d_A, M, d_B, N, L, d_inv_a, d_inv_b, d_out, ldo);
}

// GEMM panel (complete rows: the ranking pages through them) + rank, tiled over the queries; panel <= 1 GiB of float64
PVS_EXPORT int pvs_cosine_topk_f64_dev(pvs_ctx* ctx, const double* d_Q, int64_t nq, const double* d_DB, int64_t N, int64_t L,
                                       const double* d_inv_q, const double* d_inv_db, int k, int64_t* d_idx, double* d_val) {
  PVS_NEED(ctx, "ctx");
  if (nq <= 0) return PVS_OK;
  PVS_NEED(d_Q, "Q");
  PVS_NEED(d_DB, "DB");
  PVS_NEED(d_idx, "idx");
  PVS_NEED(d_val, "val");
  if (N <= 0) PVS_FAIL(PVS_ERR_INVALID, "empty database");
  if (k < 1 || k > N) PVS_FAIL(PVS_ERR_INVALID, "k = %d out of range 1..%lld", k, (long long)N);
  PVS_HIP(hipSetDevice(ctx->device));
  const int64_t QT = std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)128 << 20) / N));
  double* panel = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, (size_t)QT * N * sizeof(double), reinterpret_cast<void**>(&panel)));
  for (int64_t q0 = 0; q0 < nq; q0 += QT) {
    const int64_t qn = std::min(QT, nq - q0);
    // the whole problem in one panel and Q == DB: the symmetric kernel (launch_cosine_f64_dev detects it)
    PVS_TRY(launch_cosine_f64_dev(ctx, d_Q + q0 * L, qn, d_DB, N, L, d_inv_q ? d_inv_q + q0 : nullptr, d_inv_db, panel, N));
    PVS_TRY(launch_rank_f64(ctx, panel, qn, N, N, k, d_idx + q0 * k, d_val + q0 * k));
  }
  return PVS_OK;
}

// host pointers: the database is uploaded ONCE per call, the queries in blocks; norms, panels and lists stay on the device
PVS_EXPORT int pvs_cosine_topk_f64(pvs_ctx* ctx, const double* Q, int64_t nq, const double* DB, int64_t N, int64_t L, int k,
                                   int64_t* out_idx, double* out_val) {
  PVS_NEED(ctx, "ctx");
  if (nq <= 0) return PVS_OK;
  if (L <= 1) PVS_FAIL(PVS_ERR_INVALID, "Cosine similarity requires at least 2 features. Got %lld features.", (long long)L);
  PVS_NEED(Q, "Q");
  PVS_NEED(DB, "DB");
  PVS_NEED(out_idx, "idx");
  PVS_NEED(out_val, "val");
  if (N <= 0) PVS_FAIL(PVS_ERR_INVALID, "empty database");
  if (k < 1 || k > N) PVS_FAIL(PVS_ERR_INVALID, "k = %d out of range 1..%lld", k, (long long)N);
  PVS_HIP(hipSetDevice(ctx->device));
  const bool same = (Q == DB && nq == N);
  // query block: as many rows as one score panel of pvs_cosine_topk_f64_dev holds, at most 2 GiB of rows
  const int64_t QB = same ? nq : std::max<int64_t>(1, std::min<int64_t>(nq, std::min<int64_t>(((int64_t)128 << 20) / N,
                                                                                             ((int64_t)256 << 20) / L)));
  const size_t db_bytes = ((size_t)N * L * 8 + 255) / 256 * 256, q_bytes = same ? 0 : ((size_t)QB * L * 8 + 255) / 256 * 256;
  char* d_in = nullptr;
  char* d_s = nullptr;
  PVS_TRY(ws_reserve(ctx, 0, db_bytes + q_bytes, reinterpret_cast<void**>(&d_in)));
  const size_t nrm_b = ((size_t)(N + QB) * 8 + 255) / 256 * 256, idx_b = ((size_t)QB * k * 8 + 255) / 256 * 256;
  PVS_TRY(ws_reserve(ctx, 6, nrm_b + 2 * idx_b, reinterpret_cast<void**>(&d_s)));
  double* d_db = reinterpret_cast<double*>(d_in);
  double* d_q = same ? d_db : reinterpret_cast<double*>(d_in + db_bytes);
  double* inv_db = reinterpret_cast<double*>(d_s);
  double* inv_q = same ? inv_db : inv_db + N;
  int64_t* d_idx = reinterpret_cast<int64_t*>(d_s + nrm_b);
  double* d_val = reinterpret_cast<double*>(d_s + nrm_b + idx_b);
  PVS_HIP(hipMemcpyAsync(d_db, DB, (size_t)N * L * 8, hipMemcpyHostToDevice, ctx->stream));
  PVS_TRY(launch_row_inv_norms_f64(ctx, d_db, N, L, inv_db));
  for (int64_t q0 = 0; q0 < nq; q0 += QB) {
    const int64_t qn = std::min(QB, nq - q0);
    if (!same) {
      PVS_HIP(hipMemcpyAsync(d_q, Q + q0 * L, (size_t)qn * L * 8, hipMemcpyHostToDevice, ctx->stream));
      PVS_TRY(launch_row_inv_norms_f64(ctx, d_q, qn, L, inv_q));
    }
    PVS_TRY(pvs_cosine_topk_f64_dev(ctx, d_q, qn, d_db, N, L, inv_q, inv_db, k, d_idx, d_val));
    PVS_HIP(hipMemcpyAsync(out_idx + q0 * k, d_idx, (size_t)qn * k * 8, hipMemcpyDeviceToHost, ctx->stream));
    PVS_HIP(hipMemcpyAsync(out_val + q0 * k, d_val, (size_t)qn * k * 8, hipMemcpyDeviceToHost, ctx->stream));
    PVS_HIP(hipStreamSynchronize(ctx->stream));   // d_q and the lists are reused by the next block
  }
  return PVS_OK;
}

// ================================================================================ vocabulary training
// Through a pinned staging block of the context: a copy into pageable caller memory makes the driver pin the caller's pages on
// every call (0.5 ms for 262 KB in the HIP trace of the training loop; the results come back once per iteration).
static int stats_to_host(pvs_ctx* ctx, const double* d_stats, size_t n, double* h_out) {
  const size_t bytes = n * sizeof(double);
  if (bytes <= ((size_t)8 << 20)) {
    if (ctx->h_stage_bytes < bytes) {
      if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
      ctx->h_stage = nullptr;
      ctx->h_stage_bytes = 0;
      const size_t want = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
      if (hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault) == hipSuccess) ctx->h_stage_bytes = want;
      else ctx->h_stage = nullptr;
    }
    if (ctx->h_stage) {
      PVS_HIP(hipMemcpyAsync(ctx->h_stage, d_stats, bytes, hipMemcpyDeviceToHost, ctx->stream));
      PVS_HIP(hipStreamSynchronize(ctx->stream));
      memcpy(h_out, ctx->h_stage, bytes);
      return PVS_OK;
    }
  }
  PVS_HIP(hipMemcpyAsync(h_out, d_stats, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}

PVS_EXPORT int pvs_materialise_dev(pvs_ctx* ctx, const void* d_desc, int desc_kind, int D, int64_t total_desc, float* d_out) {
  PVS_NEED(ctx, "ctx");
  PVS_TRY(check_kind(desc_kind));
  if (total_desc < 0 || D <= 0) PVS_FAIL(PVS_ERR_INVALID, "bad sizes");
  if (total_desc == 0) return PVS_OK;
  PVS_NEED(d_desc, "descriptors");
  PVS_NEED(d_out, "out");
  PVS_HIP(hipSetDevice(ctx->device));
  return launch_materialise(ctx, d_desc, desc_kind, total_desc, D, d_out);
}

PVS_EXPORT int pvs_kmeans_step_dev(pvs_ctx* ctx, const pvs_codebook* cb, const float* d_x, int64_t total_desc, int32_t* d_labels,
                                   const int32_t* d_prev_labels, double* h_stats, float* d_sqdist) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(cb, "codebook");
  PVS_NEED(d_x, "descriptors");
  PVS_NEED(d_labels, "labels");
  PVS_NEED(h_stats, "stats");
  if (total_desc <= 0) PVS_FAIL(PVS_ERR_INVALID, "k-means needs at least one descriptor");
  PVS_HIP(hipSetDevice(ctx->device));
  const size_t n = (size_t)cb->K * cb->D + cb->K + 2;
  double* d_stats = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, n * sizeof(double), reinterpret_cast<void**>(&d_stats)));
  PVS_TRY(launch_kmeans_step(ctx, cb, d_x, total_desc, d_labels, d_prev_labels, d_stats, d_sqdist));
  return stats_to_host(ctx, d_stats, n, h_stats);
}

PVS_EXPORT int pvs_gmm_em_step_dev(pvs_ctx* ctx, const pvs_gmm* gmm, const float* d_x, int64_t total_desc, double* h_stats) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(gmm, "gmm");
  PVS_NEED(d_x, "descriptors");
  PVS_NEED(h_stats, "stats");
  if (total_desc <= 0) PVS_FAIL(PVS_ERR_INVALID, "GMM training needs at least one descriptor");
  PVS_HIP(hipSetDevice(ctx->device));
  const size_t n = (size_t)gmm->K + (size_t)2 * gmm->K * gmm->D + 1;
  double* d_stats = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, n * sizeof(double), reinterpret_cast<void**>(&d_stats)));
  PVS_TRY(launch_gmm_em_step(ctx, gmm, d_x, gmm->D, total_desc, d_stats));
  return stats_to_host(ctx, d_stats, n, h_stats);
}

PVS_EXPORT int pvs_label_sums_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, const int32_t* d_labels, int K, int square,
                                  double* h_out) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(d_x, "descriptors");
  PVS_NEED(d_labels, "labels");
  PVS_NEED(h_out, "out");
  if (total_desc <= 0 || D <= 0 || K <= 0) PVS_FAIL(PVS_ERR_INVALID, "bad sizes");
  PVS_HIP(hipSetDevice(ctx->device));
  const size_t n = (size_t)K * D;
  double* d_out = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, n * sizeof(double), reinterpret_cast<void**>(&d_out)));
  PVS_TRY(launch_label_sums(ctx, d_x, total_desc, D, d_labels, K, square, d_out));
  return stats_to_host(ctx, d_out, n, h_out);
}

PVS_EXPORT int pvs_gram_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, double* h_out) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(d_x, "descriptors");
  PVS_NEED(h_out, "out");
  if (total_desc <= 0 || D <= 0) PVS_FAIL(PVS_ERR_INVALID, "the Gram matrix needs at least one row");
  PVS_HIP(hipSetDevice(ctx->device));
  const size_t n = (size_t)D + (size_t)D * D;
  double* d_out = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, n * sizeof(double), reinterpret_cast<void**>(&d_out)));
  PVS_TRY(launch_gram(ctx, d_x, total_desc, D, d_out));
  return stats_to_host(ctx, d_out, n, h_out);
}

PVS_EXPORT int pvs_seed_distances_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, const float* cand, int n_cand,
                                      const float* d_mind, float* d_dist, double* h_pot, int cand_on_device) {
  const float* h_cand = cand;
  PVS_NEED(ctx, "ctx");
  PVS_NEED(d_x, "descriptors");
  PVS_NEED(h_cand, "candidates");
  PVS_NEED(d_dist, "dist");
  PVS_NEED(h_pot, "pot");
  if (n_cand < 1 || n_cand > 8 || D <= 0) PVS_FAIL(PVS_ERR_INVALID, "1..8 candidates of positive dimension");
  PVS_HIP(hipSetDevice(ctx->device));
  char* ws = nullptr;
  const size_t cand_b = ((size_t)n_cand * D * 4 + 255) / 256 * 256;
  PVS_TRY(ws_reserve(ctx, 2, cand_b + 8 * sizeof(double), reinterpret_cast<void**>(&ws)));
  float* d_cand = reinterpret_cast<float*>(ws);
  double* d_pot = reinterpret_cast<double*>(ws + cand_b);
  if (!cand_on_device) PVS_HIP(hipMemcpyAsync(d_cand, h_cand, (size_t)n_cand * D * 4, hipMemcpyHostToDevice, ctx->stream));
  PVS_TRY(launch_seed_distances(ctx, d_x, total_desc, D, cand_on_device ? cand : d_cand, n_cand, d_mind, d_dist, d_pot));
  double pot8[8];
  PVS_TRY(stats_to_host(ctx, d_pot, 8, pot8));
  for (int j = 0; j < n_cand; ++j) h_pot[j] = pot8[j];
  return PVS_OK;
}

PVS_EXPORT int pvs_seed_pick_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, const float* d_mind, const int64_t* h_blocks,
                                 const double* h_base, const double* h_target, int n_cand, float* d_cand, int64_t* h_idx) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(d_x, "descriptors");
  PVS_NEED(d_mind, "mind");
  PVS_NEED(h_blocks, "blocks");
  PVS_NEED(h_base, "base");
  PVS_NEED(h_target, "target");
  PVS_NEED(d_cand, "candidates");
  PVS_NEED(h_idx, "idx");
  if (n_cand < 1 || n_cand > 64 || D <= 0 || total_desc <= 0) PVS_FAIL(PVS_ERR_INVALID, "1..64 candidates");
  const int64_t nblk = (total_desc + 4095) / 4096;
  for (int c = 0; c < n_cand; ++c)
    if (h_blocks[c] < 0 || h_blocks[c] >= nblk) PVS_FAIL(PVS_ERR_INVALID, "candidate block out of range");
  PVS_HIP(hipSetDevice(ctx->device));
  char* ws = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, (size_t)n_cand * 32, reinterpret_cast<void**>(&ws)));
  int64_t* d_blk = reinterpret_cast<int64_t*>(ws);
  double* d_base = reinterpret_cast<double*>(ws + (size_t)n_cand * 8);
  double* d_tgt = reinterpret_cast<double*>(ws + (size_t)n_cand * 16);
  int64_t* d_idx = reinterpret_cast<int64_t*>(ws + (size_t)n_cand * 24);
  PVS_HIP(hipMemcpyAsync(d_blk, h_blocks, (size_t)n_cand * 8, hipMemcpyHostToDevice, ctx->stream));
  PVS_HIP(hipMemcpyAsync(d_base, h_base, (size_t)n_cand * 8, hipMemcpyHostToDevice, ctx->stream));
  PVS_HIP(hipMemcpyAsync(d_tgt, h_target, (size_t)n_cand * 8, hipMemcpyHostToDevice, ctx->stream));
  PVS_TRY(launch_seed_pick(ctx, d_x, total_desc, D, d_mind, d_blk, d_base, d_tgt, n_cand, d_idx, d_cand));
  PVS_HIP(hipMemcpyAsync(h_idx, d_idx, (size_t)n_cand * 8, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}

PVS_EXPORT int pvs_min_update_dev(pvs_ctx* ctx, float* d_mind, const float* d_dist, int64_t total_desc, double* h_block_sums) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(d_mind, "mind");
  PVS_NEED(h_block_sums, "block sums");
  if (total_desc <= 0) PVS_FAIL(PVS_ERR_INVALID, "empty input");
  PVS_HIP(hipSetDevice(ctx->device));
  const size_t nblk = (size_t)((total_desc + 4095) / 4096);
  double* d_bs = nullptr;
  PVS_TRY(ws_reserve(ctx, 2, nblk * sizeof(double), reinterpret_cast<void**>(&d_bs)));
  PVS_TRY(launch_min_update(ctx, d_mind, d_dist, total_desc, d_bs));
  return stats_to_host(ctx, d_bs, nblk, h_block_sums);
}

PVS_EXPORT int pvs_kmeanspp_run_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, int n_clusters, int trials,
                                    const double* h_uniform, int64_t* h_indices) {
  PVS_NEED(ctx, "ctx");
  PVS_NEED(d_x, "descriptors");
  PVS_NEED(h_uniform, "uniform numbers");
  PVS_NEED(h_indices, "indices");
  if (D <= 0 || total_desc <= 0 || n_clusters < 1 || n_clusters > total_desc) PVS_FAIL(PVS_ERR_INVALID, "sizes");
  if (trials < 1 || trials > 8) PVS_FAIL(PVS_ERR_UNSUPPORTED, "1..8 local trials (got %d): use the stepwise entry points", trials);
  if (h_indices[0] < 0 || h_indices[0] >= total_desc) PVS_FAIL(PVS_ERR_INVALID, "first centre out of range");
  PVS_HIP(hipSetDevice(ctx->device));
  const size_t nblk = (size_t)((total_desc + 4095) / 4096);
  auto al = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t mind_b = al((size_t)total_desc * 4), dist_b = al((size_t)trials * total_desc * 4), cand_b = al((size_t)trials * D * 4),
               bs_b = al(nblk * 8), uni_b = al((size_t)std::max(n_clusters - 1, 1) * trials * 8), idx_b = al((size_t)n_clusters * 8), small_b = 512;
  char* ws = nullptr;
  PVS_TRY(ws_reserve(ctx, 3, mind_b + dist_b + cand_b + bs_b + uni_b + idx_b + small_b, reinterpret_cast<void**>(&ws)));
  float* d_mind = reinterpret_cast<float*>(ws);
  float* d_dist = reinterpret_cast<float*>(ws + mind_b);
  float* d_cand = reinterpret_cast<float*>(ws + mind_b + dist_b);
  double* d_bs = reinterpret_cast<double*>(ws + mind_b + dist_b + cand_b);
  double* d_uni = reinterpret_cast<double*>(ws + mind_b + dist_b + cand_b + bs_b);
  int64_t* d_indices = reinterpret_cast<int64_t*>(ws + mind_b + dist_b + cand_b + bs_b + uni_b);
  char* d_small = ws + mind_b + dist_b + cand_b + bs_b + uni_b + idx_b;
  PVS_HIP(hipMemsetAsync(d_mind, 0x7f, (size_t)total_desc * 4, ctx->stream));     // 3.39e38: "no centre yet"
  if (n_clusters > 1)
    PVS_HIP(hipMemcpyAsync(d_uni, h_uniform, (size_t)(n_clusters - 1) * trials * 8, hipMemcpyHostToDevice, ctx->stream));
  PVS_HIP(hipMemcpyAsync(d_indices, h_indices, 8, hipMemcpyHostToDevice, ctx->stream));
  PVS_TRY(launch_kmeanspp_run(ctx, d_x, total_desc, D, n_clusters, trials, d_uni, d_mind, d_dist, d_cand, d_bs, d_small, d_indices));
  PVS_HIP(hipMemcpyAsync(h_indices, d_indices, (size_t)n_clusters * 8, hipMemcpyDeviceToHost, ctx->stream));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  return PVS_OK;
}

// ================================================================================ diagnostics
PVS_EXPORT int pvs_fused_profile(pvs_ctx* ctx, int enable, int64_t* out16) {
  PVS_NEED(ctx, "ctx");
  PVS_HIP(hipSetDevice(ctx->device));
  PVS_HIP(hipStreamSynchronize(ctx->stream));
  if (out16) {
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    if (ctx->d_fused_stamps) PVS_HIP(hipMemcpy(out16, ctx->d_fused_stamps, 16 * 8, hipMemcpyDeviceToHost));
  }
  if (enable) {
    if (!ctx->d_fused_stamps) PVS_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_fused_stamps), 16 * 8));
    PVS_HIP(hipMemset(ctx->d_fused_stamps, 0, 16 * 8));
  } else if (ctx->d_fused_stamps) {
    PVS_HIP(hipFree(ctx->d_fused_stamps));
    ctx->d_fused_stamps = nullptr;
  }
  return PVS_OK;
}

// ================================================================================ timers
PVS_EXPORT int pvs_timers_enable(pvs_ctx* ctx, int on) {
  PVS_NEED(ctx, "ctx");
  PVS_TRY(drain_timers(ctx));
  ctx->timers_on = on != 0;
  return PVS_OK;
}
PVS_EXPORT int pvs_timers_reset(pvs_ctx* ctx) {
  PVS_NEED(ctx, "ctx");
  PVS_TRY(drain_timers(ctx));
  for (int i = 0; i < PVS_TIMER_SLOTS; ++i) {
    ctx->t_total[i] = 0;
    ctx->t_count[i] = 0;
  }
  return PVS_OK;
}
PVS_EXPORT int pvs_timers_read(pvs_ctx* ctx, int which, double* total_ms, int64_t* launches) {
  PVS_NEED(ctx, "ctx");
  if (which < 0 || which >= PVS_TIMER_SLOTS) PVS_FAIL(PVS_ERR_INVALID, "timer slot %d out of range", which);
  PVS_TRY(drain_timers(ctx));
  if (total_ms) *total_ms = ctx->t_total[which];
  if (launches) *launches = ctx->t_count[which];
  return PVS_OK;
}
