/*
 * pvsim.h -- C-ABI of libpvsim_hip.so: the MI355X (gfx950) hot path of an image-similarity engine
 * that is a drop-in for pyvisim's VLADEncoder / FisherVectorEncoder / cosine_similarity / eval top-k.
 *
 * The reference (MechaCritter/Python-Visual-Similarity, pure Python) has no FFI layer; its extension
 * points are Python objects.  This header is therefore the NEW boundary that sits
 * directly beneath those Python methods; each entry point names the reference code it replaces
 * (paths relative to the reference root).  The Python package `pvsim` binds it with ctypes
 * (python-visual-similarity_amd/pvsim/_ffi.py); INTEGRATION.md shows the binding a pyvisim maintainer
 * would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no C++/torch types.
 *   - every function returns a pvs_status (0 = ok).  pvs_last_error() returns a thread-local message.
 *     No exception or abort crosses the ABI.
 *   - the library never owns host memory: outputs are caller-allocated, C-contiguous.
 *   - device tables live behind opaque handles with explicit create/destroy.
 *   - one pvs_ctx = one device + one HIP stream.  Entry points ending in `_dev` take DEVICE pointers
 *     (e.g. torch.Tensor.data_ptr(), or pvs_malloc memory), enqueue on the context's stream and return
 *     without synchronising; all others take HOST pointers and block until the result is in host memory.
 *   - there is no CPU fallback: without a GPU every compute entry point fails with PVS_ERR_NO_DEVICE.
 */
#ifndef PVSIM_H
#define PVSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVS_VERSION 103 /* 0.1.3 */

typedef enum {
  PVS_OK = 0,
  PVS_ERR_INVALID = 1,     /* bad argument                      -> ValueError   */
  PVS_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime error -> RuntimeError */
  PVS_ERR_OOM = 3,         /* device allocation failed          -> MemoryError  */
  PVS_ERR_UNSUPPORTED = 4, /* shape / option not implemented    -> NotImplementedError */
  PVS_ERR_DIM = 5          /* dimension mismatch                -> RuntimeError (as the reference raises) */
} pvs_status;

/* How descriptor rows are stored, and whether the RootSIFT tail
 * (pyvisim/features/_features.py:112-114: d /= sum(d)+1e-7; d = sqrt(d)) is fused into the load. */
typedef enum {
  PVS_DESC_F32 = 0,          /* float32 rows used as they are                                */
  PVS_DESC_F32_ROOTSIFT = 1, /* float32 raw SIFT rows (0..255), RootSIFT applied on the fly  */
  PVS_DESC_U8_ROOTSIFT = 2   /* uint8  raw SIFT rows, RootSIFT applied on the fly (4x fewer HBM bytes) */
} pvs_desc_kind;

/* Behaviour switches of a context (pvs_set_option).  The defaults are the product path; the other values exist so that
 * tests and benchmarks can pin one of several implementations that must agree bit for bit. */
typedef enum {
  PVS_OPT_ASSIGN_PREFILTER = 0, /* 1 (default): fp16 MFMA prefilter (three products) + exact pass on near ties; 0: exact f32 MFMA kernel   */
                                /* only; 2 / 3: measurement variants with two / one fp16 product(s) and the wider margin that goes with them; */
                                /* 4: three products on v_mfma_f32_16x16x32_f16 (D = 128, 128 < K <= 256: measurement variant, same lists)   */
  PVS_OPT_VLAD_PATH = 1,        /* 0 (default) and 1: assign + gather aggregate (two reads of the descriptors); 2: assign +        */
                                /* streaming aggregate; 3: fused one-read kernel (D = 128, 128 < K <= 256; error otherwise)        */
  PVS_OPT_TOPK_SELECT_ONLY = 2, /* top-k kernel for k <= 16: 0 (default) threshold filter on panels of >= 4096 columns, k rounds         */
                                /* otherwise; 1: always the radix-select kernel; 2: always the rounds; 3: always the threshold filter  */
  PVS_OPT_AGG_VARIANT = 3,      /* gather aggregate at D <= 128: 0 (default) chosen by rows per cluster; 1: eight waves per SIMD,     */
                                /* batches of 4 rows (short images); 2: five waves, batches of 8 (long images).  Same bits.         */
  PVS_OPT_FISHER_SCALE = 4,     /* division of a Fisher row by its norm: 0 (default) and 1: a second pass over the rows; 2: inside the      */
                                /* moments kernel (one workgroup per image; 1.6 % less time, 1.7 x the bytes beyond L2 at configs[2]).     */
                                /* Same bits.                                                                                             */
  PVS_OPT_COUNT_ = 5
} pvs_option;

typedef struct pvs_ctx pvs_ctx;
typedef struct pvs_codebook pvs_codebook; /* KMeans.cluster_centers_ (K,D) f32 + ||c||^2                */
typedef struct pvs_gmm pvs_gmm;           /* GaussianMixture weights_/means_/covariances_ ('diag')     */
typedef struct pvs_pca pvs_pca;           /* PCA components_ (C,Din) + mean_                            */
typedef struct pvs_comm pvs_comm;         /* one rank of the multi-GPU exchange (RCCL communicator bound to a context)  */

/* Normalisation knobs shared by both encoders -- the constructor kwargs of
 * VLADEncoder (pyvisim/encoders/vlad.py:42-53) and FisherVectorEncoder (fisher_vector.py:41-51). */
typedef struct {
  double power_norm_weight; /* p in sign(v)|v|^p ; VLAD default 1, Fisher default 0.5 */
  double norm_order;        /* ord of np.linalg.norm: 1, 2, any p > 0, or +INFINITY   */
  double epsilon;           /* added to the norm before dividing (default 1e-9)       */
} pvs_norm_params;

/* ---------------------------------------------------------------- context / errors */
int pvs_version(void);
const char* pvs_last_error(void);
int pvs_device_count(int* count);
/* stream == NULL: the context creates and owns a stream.  Otherwise `stream` is a hipStream_t the
 * caller owns (e.g. torch.cuda.current_stream().cuda_stream) and all work is enqueued there. */
int pvs_init(int device_id, void* stream, pvs_ctx** out);
int pvs_destroy(pvs_ctx* ctx);
int pvs_sync(pvs_ctx* ctx);
void* pvs_stream(pvs_ctx* ctx);
int pvs_device_name(pvs_ctx* ctx, char* buf, size_t buflen);
int pvs_set_option(pvs_ctx* ctx, int option, int value);
int pvs_get_option(pvs_ctx* ctx, int option, int* value);

/* plain device memory for hosts that do not use torch */
int pvs_malloc(pvs_ctx* ctx, size_t bytes, void** dptr);
int pvs_free(pvs_ctx* ctx, void* dptr);
int pvs_memcpy_h2d(pvs_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int pvs_memcpy_d2h(pvs_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int pvs_memset(pvs_ctx* ctx, void* dst_dev, int value, size_t bytes);

/* 4- or 8-byte pattern fill of device memory on the context's stream (list buffers: -1 indices, -inf scores). */
int pvs_fill_dev(pvs_ctx* ctx, void* dst_dev, int64_t n_elems, int elem_bytes, uint64_t pattern);
/* Stream ordering between two contexts of one process (e.g. compute and exchange): work queued on `waiter` after this
 * call starts only when everything queued on `signal` before this call has finished.  No host synchronisation. */
int pvs_stream_wait(pvs_ctx* waiter, pvs_ctx* signal);

/* ---------------------------------------------------------------- tables (host pointers in) */
/* replaces reading KMeans.cluster_centers_ per image (vlad.py:96) */
int pvs_codebook_create(pvs_ctx* ctx, const float* centroids /*[K*D]*/, int K, int D, pvs_codebook** out);
int pvs_codebook_destroy(pvs_ctx* ctx, pvs_codebook* cb);
/* replaces reading weights_/means_/covariances_ per image (fisher_vector.py:95-97); fp64 in, as sklearn stores */
int pvs_gmm_create(pvs_ctx* ctx, const double* weights /*[K]*/, const double* means /*[K*D]*/,
                   const double* covariances /*[K*D]*/, int K, int D, pvs_gmm** out);
int pvs_gmm_destroy(pvs_ctx* ctx, pvs_gmm* g);
/* replaces PCA.transform (vlad.py:89-90, fisher_vector.py:91-92) */
int pvs_pca_create(pvs_ctx* ctx, const float* components /*[C*Din]*/, const float* mean /*[Din]*/,
                   int n_components, int d_in, pvs_pca** out);
int pvs_pca_destroy(pvs_ctx* ctx, pvs_pca* p);

/* ---------------------------------------------------------------- VLAD: vlad.py:81-115
 * desc: packed rows of all images [offsets[n_images]][D_in]; offsets: int64[n_images+1] (CSR style).
 * pca may be NULL.  out: float32 [n_images][K*D] (k-major, flatten=True layout).  An image with zero
 * descriptors yields a zero row (the reference aborts the batch, vlad.py:92-93 -- fenced quirk).
 * out_labels (optional): int32 [total descriptors], the KMeans.predict labels (vlad.py:95).
 * out_inv_norm (optional, _dev only): float32 [n_images], 1/||row||_2 (1 for zero rows) for the cosine step.
 * The host forms validate the offsets (offsets[0] == 0, non-decreasing).  pvs_vlad_encode_dev never synchronises, so it
 * cannot: d_offsets[0] == 0, non-decreasing, d_offsets[n_images] == total_desc are PRECONDITIONS of that entry point
 * (pvs_fisher_encode_dev copies the offsets to the host for batching and does check them). */
int pvs_vlad_encode(pvs_ctx* ctx, const pvs_codebook* cb, const pvs_pca* pca, const void* desc,
                    int desc_kind, const int64_t* offsets, int64_t n_images, const pvs_norm_params* prm,
                    float* out, int32_t* out_labels);
int pvs_vlad_encode_dev(pvs_ctx* ctx, const pvs_codebook* cb, const pvs_pca* pca, const void* d_desc,
                        int desc_kind, const int64_t* d_offsets, int64_t n_images, int64_t total_desc,
                        const pvs_norm_params* prm, float* d_out, int32_t* d_labels, float* d_inv_norm);
/* KMeans.predict alone (vlad.py:95 -> sklearn _k_means_lloyd.pyx:168-218) */
int pvs_kmeans_predict_dev(pvs_ctx* ctx, const pvs_codebook* cb, const void* d_desc, int desc_kind,
                           int64_t total_desc, int32_t* d_labels);

/* ---------------------------------------------------------------- Fisher: fisher_vector.py:83-135
 * out: [n_images][K + 2*K*D]  laid out [d_pi | d_mu (k-major) | d_sigma].  The host form returns
 * float64 (the reference's dtype); the device form writes float32 or float64 (out_f64 != 0). */
int pvs_fisher_encode(pvs_ctx* ctx, const pvs_gmm* g, const pvs_pca* pca, const void* desc, int desc_kind,
                      const int64_t* offsets, int64_t n_images, const pvs_norm_params* prm, double* out);
int pvs_fisher_encode_dev(pvs_ctx* ctx, const pvs_gmm* g, const pvs_pca* pca, const void* d_desc,
                          int desc_kind, const int64_t* d_offsets, int64_t n_images, int64_t total_desc,
                          const pvs_norm_params* prm, void* d_out, int out_f64);
/* GaussianMixture.predict_proba alone (fisher_vector.py:99), float64 [total][K] */
int pvs_gmm_predict_proba_dev(pvs_ctx* ctx, const pvs_gmm* g, const void* d_desc, int desc_kind,
                              int64_t total_desc, double* d_resp);
/* PCA.transform alone: float32 [total][C] */
int pvs_pca_transform_dev(pvs_ctx* ctx, const pvs_pca* p, const void* d_desc, int desc_kind,
                          int64_t total_desc, float* d_out);

/* ---------------------------------------------------------------- cosine: pyvisim/_utils.py:312-330
 * (-> sklearn.metrics.pairwise.cosine_similarity): rows L2-normalised (zero rows stay zero), A.B^T.
 * is_f64 != 0: operands and output are float64 (the reference's dtype rule: fp32 iff both fp32). */
int pvs_cosine(pvs_ctx* ctx, const void* A, int64_t M, const void* B, int64_t N, int64_t L, int is_f64,
               void* out /*[M*N]*/);
int pvs_row_inv_norms_dev(pvs_ctx* ctx, const float* d_x, int64_t rows, int64_t L, float* d_inv);
/* out[i*ldo + j] = (A_i . B_j) * inv_a[i] * inv_b[j]; inv_* may be NULL (treated as 1). */
int pvs_cosine_dev(pvs_ctx* ctx, const float* d_A, int64_t M, const float* d_B, int64_t N, int64_t L,
                   const float* d_inv_a, const float* d_inv_b, float* d_out, int64_t ldo);

/* Same, and additionally the transposed panel out_t[j*ldt + i] = out[i*ldo + j] (bit-identical): one GEMM then serves
 * the row queries AND the column queries of a block pair (symmetric multi-GPU scheme, pvsim/distributed.py). */
int pvs_cosine_dual_dev(pvs_ctx* ctx, const float* d_A, int64_t M, const float* d_B, int64_t N, int64_t L,
                        const float* d_inv_a, const float* d_inv_b, float* d_out, int64_t ldo, float* d_out_t, int64_t ldt);

/* ---------------------------------------------------------------- top-k: pyvisim/eval.py:37-43,75-80,131-132
 * per query row: np.argsort(-scores)[:k].  Order is (score desc, index asc); NaN scores rank last.
 * pvs_topk_dev consumes a score panel [nq][ncols] (row stride ld) whose column 0 has global index
 * col_offset; with merge != 0 the panel is merged into the running lists already in d_idx/d_val.
 * d_idx: int64 [nq*k], d_val: float32 [nq*k]. */
int pvs_topk_dev(pvs_ctx* ctx, const float* d_scores, int64_t nq, int64_t ncols, int64_t ld, int k,
                 int64_t col_offset, int merge, int64_t* d_idx, float* d_val);
/* cosine + top-k without materialising the full nq x N matrix (tiled through a workspace panel). */
int pvs_cosine_topk_dev(pvs_ctx* ctx, const float* d_Q, int64_t nq, const float* d_DB, int64_t N, int64_t L,
                        const float* d_inv_q, const float* d_inv_db, int k, int64_t col_offset, int merge,
                        int64_t* d_idx, float* d_val);
/* fp16 operands (BASELINE configs[4]: the 1M x 1M similarity on fp16 encodings): fp32 rows are converted once
 * (round-to-nearest-even; keep the intra-normalised VLAD values and pass the fp32 1/||row|| factors, do not
 * pre-divide -- that would push elements into fp16 subnormals), then v_mfma_f32_32x32x16_f16 with fp32 accumulate. */
int pvs_f32_to_f16_dev(pvs_ctx* ctx, const float* d_src, int64_t n, void* d_dst_f16);
int pvs_cosine_f16_dev(pvs_ctx* ctx, const void* d_A16, int64_t M, const void* d_B16, int64_t N, int64_t L,
                       const float* d_inv_a, const float* d_inv_b, float* d_out, int64_t ldo);
int pvs_cosine_topk_f16_dev(pvs_ctx* ctx, const void* d_Q16, int64_t nq, const void* d_DB16, int64_t N, int64_t L,
                            const float* d_inv_q, const float* d_inv_db, int k, int64_t col_offset, int merge,
                            int64_t* d_idx, float* d_val);
int pvs_cosine_topk(pvs_ctx* ctx, const float* Q, int64_t nq, const float* DB, int64_t N, int64_t L, int k,
                    int64_t* out_idx, float* out_val);
/* float64 operands (Fisher encodings): the reference scores and ranks in float64 unless BOTH operands are float32
 * (pyvisim/_utils.py:312-330 -> sklearn cosine_similarity; eval.py:37-43,75-80,131-132 argsort that array).  The GEMM runs
 * on the f64 matrix pipe (v_mfma_f64_16x16x4_f64, fixed k order, upper triangle + mirror when Q == DB); rows must be 16-B
 * aligned with an even L for that path (anything else takes a vector-ALU tile kernel).  out_val float64 [nq][k], same order
 * as the fp32 entry points (score descending, index ascending, NaN last); any 1 <= k <= N (deep rankings page through the
 * complete score rows, as top_k_map(k=None) needs).
 * pvs_cosine_topk_f64: host pointers, the database is uploaded once per call.  The _dev forms take device pointers and the
 * 1/||row|| factors (pvs_row_inv_norms_f64_dev; NULL = 1), keep everything resident and do not synchronise. */
int pvs_cosine_topk_f64(pvs_ctx* ctx, const double* Q, int64_t nq, const double* DB, int64_t N, int64_t L, int k,
                        int64_t* out_idx, double* out_val);
int pvs_row_inv_norms_f64_dev(pvs_ctx* ctx, const double* d_x, int64_t rows, int64_t L, double* d_inv);
int pvs_cosine_f64_dev(pvs_ctx* ctx, const double* d_A, int64_t M, const double* d_B, int64_t N, int64_t L,
                       const double* d_inv_a, const double* d_inv_b, double* d_out, int64_t ldo);
int pvs_cosine_topk_f64_dev(pvs_ctx* ctx, const double* d_Q, int64_t nq, const double* d_DB, int64_t N, int64_t L,
                            const double* d_inv_q, const double* d_inv_db, int k, int64_t* d_idx, double* d_val);
/* The same lists as pvs_cosine_topk_dev(col_offset 0, merge 0) -- bit-identical indices AND scores -- computed faster:
 * all pairs are scored with fp16 operands under a proven error bound, the columns within twice that bound of each query's
 * approximate k-th best are re-scored with the exact fp32 recurrence of the f32 GEMM kernel, and those are ranked.
 * Inputs that do not qualify (k > 128, rows not 16-B aligned or L % 8 != 0, non-finite values, or scores so crowded that more
 * than a tenth of the queries overflow their candidate slots) silently take the plain exact path.  h_stats (optional, int64[4]): [0] 1 if the filter ran, [1] queries redone by the exact path
 * (more candidates than slots), [2] candidates re-scored, [3] candidate slots per query. */
int pvs_cosine_topk_filtered_dev(pvs_ctx* ctx, const float* d_Q, int64_t nq, const float* d_DB, int64_t N, int64_t L,
                                 const float* d_inv_q, const float* d_inv_db, int k, int64_t* d_idx, float* d_val,
                                 int64_t* h_stats);
/* merges per-rank top-k lists (multi-GPU: each rank scored its own DB shard): lists [n_lists][nq][k]. */
int pvs_topk_merge_dev(pvs_ctx* ctx, const int64_t* d_idx_lists, const float* d_val_lists, int n_lists,
                       int64_t nq, int k, int64_t* d_idx, float* d_val);

/* ---------------------------------------------------------------- multi-GPU exchange (one process per GPU, RCCL over xGMI)
 * The path shards by image (images are independent: pyvisim/encoders/vlad.py:87-113; global index = block offset + local
 * index, the insertion order of pyvisim/eval.py:28); the pairwise step needs ONE exchange of the encoding blocks, plus one
 * all-to-all of k-candidate lists in the symmetric block-pair scheme (pvsim/distributed.py).  librccl is resolved at
 * pvs_comm_init (an image already mapped in the process is preferred); collectives are enqueued on the context's stream.
 * Bootstrap: rank 0 calls pvs_comm_unique_id and hands the 128 bytes to the other ranks by any host channel. */
#define PVS_UNIQUE_ID_BYTES 128
int pvs_comm_unique_id(void* out_id /*[128]*/);
int pvs_comm_init(pvs_ctx* ctx, int nranks, int rank, const void* unique_id /*[128]*/, pvs_comm** out);
int pvs_comm_destroy(pvs_comm* comm);
const char* pvs_comm_library(void);   /* which librccl image was bound ("" before the first pvs_comm_init) */
/* recv[r * bytes_per_rank ...] = rank r's send block, for every r */
int pvs_allgather_dev(pvs_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank);
/* recv[p * bytes_per_rank ...] = the block rank p addressed to this rank (its send[this rank]) */
int pvs_alltoall_dev(pvs_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank);
/* batched point-to-point: op i sends send_bytes[i] bytes to and receives recv_bytes[i] bytes from peers[i] (either may be 0) */
int pvs_sendrecv_dev(pvs_comm* comm, int n_ops, const int* peers, const void* const* d_send, const size_t* send_bytes,
                     void* const* d_recv, const size_t* recv_bytes);
/* element-wise maximum over the ranks of up to 64 host doubles (timing: slowest rank); synchronises the stream */
int pvs_allreduce_max_f64(pvs_comm* comm, double* h_inout, int count);
int pvs_comm_barrier(pvs_comm* comm);

/* ---------------------------------------------------------------- vocabulary training
 * Replaces the sklearn fits inside ImageEncoderBase.learn (pyvisim/encoders/_base_encoder.py:311-342: PCA.fit ->
 * KMeans.fit for VLAD / GaussianMixture(covariance_type="diag").fit for Fisher).  Each entry is ONE pass over the
 * stacked descriptors on the device; the K x D sized parameter update and the loop control stay with the caller
 * (pvsim/learn.py).  d_x are plain fp32 rows (n, D), ld = D: make them with pvs_materialise_dev. */
/* any descriptor kind -> plain fp32 rows (RootSIFT applied for the *_ROOTSIFT kinds). */
int pvs_materialise_dev(pvs_ctx* ctx, const void* d_desc, int desc_kind, int D, int64_t total_desc, float* d_out);
/* One Lloyd pass (sklearn/cluster/_kmeans.py:_kmeans_single_lloyd body): labels as KMeans.predict with the centres of
 * `cb`; h_stats[K*D + K + 2] = [sum over members of (x - c_k) (K*D) | member counts (K) | inertia | number of labels that
 * differ from d_prev_labels (0 if NULL)].  d_sqdist (optional, f32[n]) receives |x_i - c_label|^2. */
int pvs_kmeans_step_dev(pvs_ctx* ctx, const pvs_codebook* cb, const float* d_x, int64_t total_desc, int32_t* d_labels,
                        const int32_t* d_prev_labels, double* h_stats, float* d_sqdist);
/* One EM pass (sklearn/mixture/_base.py:_e_step + _gaussian_mixture.py:_estimate_gaussian_parameters, fp64):
 * h_stats[K + 2*K*D + 1] = [sum_i gamma_ik (K) | per k: sum_i gamma_ik x_i (D), sum_i gamma_ik x_i**2 (D) | sum_i log p(x_i)].
 * K <= 256. */
int pvs_gmm_em_step_dev(pvs_ctx* ctx, const pvs_gmm* gmm, const float* d_x, int64_t total_desc, double* h_stats);
/* h_out[K*D]: per label k the sum of x_i (square = 0) or of x_i**2 squared in fp32 (square = 1) over the descriptors with
 * d_labels[i] == k -- the hard-assignment moments GaussianMixture starts from (sklearn/mixture/_base.py, init_params="kmeans"). */
int pvs_label_sums_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, const int32_t* d_labels, int K, int square,
                       double* h_out);
/* h_out[D + D*D] = [sum_i x_i | sum_i x_i x_i^T] in fp64 (sklearn/decomposition/_pca.py covariance_eigh solver input). */
int pvs_gram_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, double* h_out);
/* k-means++ seeding (sklearn/cluster/_kmeans.py:_kmeans_plusplus): for n_cand <= 8 candidate centres (host, f32[n_cand][D])
 * d_dist[j][i] = |x_i - cand_j|^2 and h_pot[j] = sum_i min(d_mind[i], d_dist[j][i]) (d_mind NULL = +inf).  cand_on_device != 0:
 * `cand` is a device pointer (e.g. pvs_seed_pick_dev's d_cand). */
int pvs_seed_distances_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, const float* cand, int n_cand,
                           const float* d_mind, float* d_dist, double* h_pot, int cand_on_device);
/* The candidate draw of a seeding step on the device: candidate c is the first position of block h_blocks[c] (4096 entries of
 * d_mind) where the running fp64 sum, started at h_base[c], reaches h_target[c] -- searchsorted(cumsum(d_mind), r) with the
 * block chosen by the caller from pvs_min_update_dev's block sums.  Writes the rows to d_cand[c] and the indices to h_idx. */
int pvs_seed_pick_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, const float* d_mind, const int64_t* h_blocks,
                      const double* h_base, const double* h_target, int n_cand, float* d_cand, int64_t* h_idx);
/* d_mind = min(d_mind, d_dist) (d_dist NULL = keep) and h_block_sums[ceil(n/4096)] = fp64 sums of d_mind per 4096 entries. */
int pvs_min_update_dev(pvs_ctx* ctx, float* d_mind, const float* d_dist, int64_t total_desc, double* h_block_sums);

/* The whole greedy k-means++ run (sklearn/cluster/_kmeans.py:_kmeans_plusplus) from a given first centre, without a host round
 * trip per step: for c = 1 .. n_clusters-1 the targets u * pot, the candidate draws (pvs_seed_pick_dev's arithmetic, the block of a
 * draw found on the device), the candidates' distances and potentials, the best candidate and the running-minimum update are
 * enqueued back to back; one synchronisation at the end.  trials <= 8 (2 + log K for K < 404).  h_uniform: (n_clusters-1) x trials
 * numbers in [0, 1) from the caller's random stream, in draw order.  h_indices[n_clusters]: [0] = the first centre (in), the
 * others out.  The same indices as the stepwise entry points give with the same numbers. */
int pvs_kmeanspp_run_dev(pvs_ctx* ctx, const float* d_x, int D, int64_t total_desc, int n_clusters, int trials,
                         const double* h_uniform, int64_t* h_indices);

/* ---------------------------------------------------------------- measurement hooks (bench.py) */
/* Enable per-kernel-family HIP-event timing on the context's stream. which: 0 assign, 1 aggregate,
 * 2 cosine gemm, 3 top-k, 4 fisher posterior, 5 fisher moments, 6 norms/misc, 7 exact re-scoring (filtered top-k). */
#define PVS_TIMER_SLOTS 8
int pvs_timers_enable(pvs_ctx* ctx, int on);
int pvs_timers_reset(pvs_ctx* ctx);
int pvs_timers_read(pvs_ctx* ctx, int which, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* PVSIM_H */
