// GEMM descriptor shared by the engine and the C-ABI (see gemm.hip).
#pragma once
#include "b2s_common.h"

// One operand of C = op(A) * op(B).  "Stored rows" R are the non-contiguous dimension, "stored
// cols" C the contiguous one.  Non-transposed A is [M rows][K cols]; transposed A is [K rows][M cols].
// Non-transposed B is [N rows][K cols] (i.e. the Linear weight layout); transposed B is [K rows][N cols].
struct GemmOperand {
    const void* p = nullptr;
    long bs_o = 0, bs_i = 0;      // batch strides in elements (outer, inner batch index)
    int ld = 0;                   // leading dimension in elements
    int R = 0, C = 0;             // logical bounds of stored rows / cols
    // conv gather: when g_cin > 0 the operand is the virtual matrix Xg[token m][j*cin + ci] =
    // x[m + j - 2][ci] if 0 <= t+j-2 < min(len[b], T) else 0, with x stored [B*T][cin].
    int g_cin = 0, g_T = 0;
    const int* g_len = nullptr;
};

struct GemmEpilogue {
    float alpha = 1.f;
    const float* bias = nullptr;          // per output column
    int relu = 0;
    const void* relu_aux = nullptr;       // T*: out = aux > 0 ? v * aux_scale : 0   (ReLU+dropout backward)
    int ld_aux = 0;
    float aux_scale = 1.f;
    DropCfg drop = {0, 0, 1.f};           // dropout on the result, element index (z*M + m)*N + n
    const int* drop_salt = nullptr;       // optional device int mixed into the dropout key (decode step index under hipGraph replay)
    const float* residual = nullptr;      // fp32, added after dropout
    int ldr = 0;
    const int* row_len = nullptr;         // zero rows with (m % rows_per_batch) >= row_len[m / rows_per_batch]
    int rows_per_batch = 1;
    int conv_dw_cin = 0;                  // >0: output column n = j*cin+ci is stored at ci*5 + j
    int accumulate = 0;                   // fp32 output only: C += v
    float* colstat = nullptr;             // [2 N] fp32, zeroed by the caller: colstat[n] += sum_m v[m][n], colstat[N + n] += sum_m v[m][n]^2 of the
                                          // raw products (before any other epilogue step) -- the BatchNorm batch statistics of a conv layer's output
    // decode-step fusion, honoured by the weight-streaming kernel only (gemm_skinny.hip; the tiled kernels reject it): output
    // columns [kv_D, 2 kv_D) / [2 kv_D, 3 kv_D) are also appended to the head-major caches kv_k / kv_v [M][kv_D/kv_dh][kv_maxT][kv_dh] at position *kv_t
    void *kv_k = nullptr, *kv_v = nullptr;
    const int* kv_t = nullptr;
    int kv_maxT = 0, kv_D = 0, kv_dh = 0;
};

struct GemmArgs {
    GemmOperand A, B;
    int M = 0, N = 0, K = 0;
    int batch = 1, batch_inner = 1;       // z -> (z / batch_inner, z % batch_inner)
    int splitk = 1;                       // >1: K is split over blockIdx.z, partial sums atomically added (fp32 C, accumulate)
    void* C = nullptr;
    int c_fp32 = 0;                       // 1: float*, 0: compute type T*
    int ldc = 0;
    long cs_o = 0, cs_i = 0;
    GemmEpilogue epi;
    // split-K slab workspace ([splitk][M][N] fp32) owned by the CALLER (the model keeps one per stream it launches on; there
    // is no process-wide slab).  Null / too small: the partial tiles are accumulated with fp32 atomics instead.
    float* ws = nullptr;
    size_t ws_floats = 0;
};

// dtype: 0 = fp32 (mfma_f32_16x16x4f32), 1 = bf16 (mfma_f32_16x16x32_bf16, fp32 accumulate)
int b2s_gemm_launch(const GemmArgs& g, int dtype, bool ta, bool tb, hipStream_t stream);
// bf16 LDS-DMA (global_load_lds) + swizzled-LDS main loop (gemm_glds.hip)
int b2s_gemm_glds_launch(const GemmArgs& g, bool ta, bool tb, hipStream_t stream);
// decode-step weight-streaming kernel (gemm_skinny.hip); returns -1 when the problem does not fit it
int b2s_gemm_skinny_launch(const GemmArgs& g, int dtype, hipStream_t stream);
// 256x128-tile variant for the large-M forms (gemm_glds256.hip) and the split-K slab reduction shared by both
long b2s_gemm_glds256_tiles(const GemmArgs& g);
int b2s_gemm_glds256_launch(const GemmArgs& g, bool ta, bool tb, const bf16_t* zero, hipStream_t stream);
int b2s_splitk_reduce_launch(const float* ws, float* dst, int M, int N, int ldc, int splitk, int conv_dw_cin, hipStream_t stream);
// up to B2S_MAX_GROUP weight-gradient problems (TN form, fp32 accumulate, no split-K) in one launch
#define B2S_MAX_GROUP 8
struct b2s_gemm_group { int n; int order; int tile0[B2S_MAX_GROUP + 1]; GemmArgs p[B2S_MAX_GROUP]; };      // order: 1 = XCD-contiguous over the whole list
int b2s_gemm_glds256_grouped_launch(const GemmArgs* probs, int n, const bf16_t* zero, hipStream_t stream);
// 256 zero bytes in device memory, written once at first use and immutable afterwards (source of the out-of-bounds chunks of
// the LDS-DMA loads); the only process-wide device object of the GEMM layer
const bf16_t* b2s_gemm_zero_page();
int b2s_gemm_grouped_launch(const GemmArgs* probs, int n, hipStream_t stream);      // gemm.hip: + optional timing record
