/*
 * lora_hip.h — C-ABI of liblora_hip.so, the gfx950 (MI355X) implementation of the
 * LoRA fine-tuning hot path of levayz/diffusion_finetuning (`lora_diffusion`).
 *
 * The reference has no native code: every entry point below replaces a group of
 * stock-PyTorch calls made by the reference's Python.  Each declaration cites the
 * reference lines it stands in for (paths relative to the reference repo root).
 *
 * Conventions (all entry points)
 *   - plain C: raw DEVICE pointers, sizes, a `void* stream` (hipStream_t; NULL = null stream).
 *   - row-major, densely packed tensors unless a leading dimension is given.
 *   - nothing is allocated, retained or freed by the library; kernels are enqueued on
 *     `stream` and the call returns without synchronising.  Re-entrant, no mutable globals
 *     (the optional launch profiler below is the one exception and is off by default).
 *   - return value: LORA_OK (0) or a negative LORA_E_* code; never throws, never aborts.
 *   - `dtype` selects the storage/compute type of the big operands (X, W, Y, dY, dX, pred ...):
 *     LORA_F32 / LORA_F16 / LORA_BF16.  Accumulation is always fp32.  The rank-r factors,
 *     their gradients and the saved rank-r activations are ALWAYS fp32 ("master" precision).
 *
 * Notation: X[M,K] input rows, W[N,K] frozen base weight, b[N] bias, A[r,K] = lora_down.weight,
 * B[N,r] = lora_up.weight, s = LoraInjectedLinear.scale, T[M,r] = X·Aᵀ.
 */
#ifndef LORA_HIP_H
#define LORA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LORA_HIP_ABI_VERSION 9

enum lora_dtype { LORA_F32 = 0, LORA_F16 = 1, LORA_BF16 = 2 };

enum lora_status {
    LORA_OK = 0,
    LORA_E_BADARG = -1,      /* null pointer / non-positive size / unknown dtype            */
    LORA_E_RANK = -2,        /* r < 1 or r > min(K,N)  (reference: ValueError, lora.py:36-39) */
    LORA_E_ALIGN = -3,       /* pointer not 16-byte aligned where the kernel needs it        */
    LORA_E_LAUNCH = -4,      /* hipLaunch / hipGetLastError reported a failure               */
    LORA_E_UNSUPPORTED = -5  /* combination not implemented                                  */
};

/* ABI version of the loaded library (== LORA_HIP_ABI_VERSION it was built with). */
int lora_version(void);

/* Static, human-readable text for a lora_status code. */
const char* lora_status_string(int status);

/*
 * Packed rank-r factors: the compute-dtype copies of lora_down / lora_up the fused kernels stream, rank padded
 * to 16 with zeros, BOTH orientations per factor so that each is a plain row-major operand tile:
 *     Apack (32·K elements) = [ A16 [16,K] : A16[j,k]  = A[j,k] | At16[K,16] : At16[k,j] = A[j,k] ]
 *     Bpack (32·N elements) = [ Bt16[16,N] : Bt16[j,n] = B[n,j] | B16 [N,16] : B16[n,j]  = B[n,j] ]
 * The [16,len] halves are the main-loop factors (forward T = X·Aᵀ uses A16, backward U = dY·B uses Bt16), the
 * [len,16] halves the epilogue factors (forward B16, backward At16).  This is the same cast of
 * lora_down / lora_up to the compute dtype that autocast applies on lora.py:50 in the reference.  Re-pack
 * whenever A or B changed (once per optimizer step).  The batched form packs every layer of a flat parameter
 * slab in ONE launch: table[l] = {a_off, b_off, K, N, r, apack_off, bpack_off, 0} (int64, device memory;
 * element offsets into `params` resp. `packed`), max_len = max over layers of max(K,N).
 * For r > 16 nothing is packed (those ranks run on the shape-agnostic kernels that read the fp32 masters).
 */
int lora_pack_factors(const float* A, const float* B, void* Apack, void* Bpack, int K, int N, int r,
                      int dtype, void* stream);
int lora_pack_factors_batched(const int64_t* table, int n_layers, int max_len, const float* params,
                              void* packed, int dtype, void* stream);

/* Packing with explicit destinations, one table row per FACTOR (grouped layers place several layers' factors in one
 * operand): table[i] = {src_off, which (0: A [r,len], 1: B [len,r]), len, r, d16_off, d16_ld, dT_off, rows} (int64,
 * device).  Rows j < rows of the [16,len] form go to packed[d16_off + j·d16_ld + c], columns j < rows of the [len,16]
 * form to packed[dT_off + c·16 + j]; an offset of -1 skips a form; rows = 16 zero-fills unused rank slots, rows = r
 * leaves them alone (block-diagonal groups write only their own slots of a buffer zeroed once). */
int lora_pack_items(const int64_t* table, int n_items, int max_len, const float* params, void* packed, int dtype,
                    void* stream);

/*
 * Forward of LoraInjectedLinear.forward — lora_diffusion/lora.py:49-50
 *     Y = X·Wᵀ + b + s·((X·Aᵀ)·Bᵀ)
 * replaces F.linear ×3 + mul + add (5 launches + the [M,N] LoRA temporary) with one kernel.
 * A, B: fp32 masters; Apack, Bpack: their packed forms (both nullable → shape-agnostic slow path).
 * T_out receives T = X·Aᵀ [M,r] fp32 for the backward.
 */
int lora_linear_fwd(const void* X, const void* W, const void* bias /* nullable */,
                    const float* A, const float* B, const void* Apack, const void* Bpack, void* Y,
                    float* T_out, int64_t M, int K, int N, int r, float scale, int dtype,
                    void* stream);
/* The same with optional split-K scratch (see lora_linear_bwd_input_ws / lora_gemm_workspace_bytes(M, K, N, dtype)):
 * the 1280-wide projections at 1024 and 256 rows leave a third of the chip idle unless their contraction is cut. */
int lora_linear_fwd_ws(const void* X, const void* W, const void* bias /* nullable */,
                       const float* A, const float* B, const void* Apack, const void* Bpack, void* Y,
                       float* T_out, int64_t M, int K, int N, int r, float scale, int dtype,
                       void* workspace /* nullable */, int64_t ws_bytes, void* stream);

/*
 * The same forward with the GEGLU gate of its caller folded into the epilogue — diffusers GEGLU.forward
 *     hidden, gate = proj(x).chunk(2, dim=-1);  return hidden * gelu(gate)
 * where `proj` is a LoraInjectedLinear (target class "GEGLU", lora_diffusion/lora.py:53):
 *     Y = X·Wᵀ + b + s·((X·Aᵀ)·Bᵀ)  [M,N],   Out = Y[:, :N/2] · gelu(Y[:, N/2:])  [M,N/2]   (exact gelu, fp32 math on the
 *     values of Y rounded to `dtype`, exactly what geglu_gate_fwd computes from a stored Y)
 * in ONE launch: a column tile owns 64 hidden columns and the 64 gate columns behind them.  Y may be NULL (inference /
 * the no-grad pass of gradient checkpointing): then the [M,N] activation is never written.  16-bit dtypes, packed factors,
 * K % 64 == 0 and N % 128 == 0 only — anything else returns LORA_E_UNSUPPORTED and the caller runs lora_linear_fwd followed
 * by geglu_gate_fwd.
 */
int lora_linear_geglu_fwd(const void* X, const void* W, const void* bias /* nullable */, const void* Apack,
                          const void* Bpack, void* Y /* nullable */, void* Out, float* T_out, int64_t M, int K, int N,
                          int r, float scale, int dtype, void* stream);

/*
 * Backward of the same gate, in the epilogue of the GEMM that produces its incoming gradient.  In a transformer block the
 * gated activation feeds a frozen linear layer (diffusers FeedForward: net = [GEGLU, Dropout, Linear]); autograd of
 *     z = (hidden * gelu(gate)) @ W2ᵀ + b2 ,   [hidden | gate] = Y = proj(x)      (proj: the LoraInjectedLinear above)
 * needs  dout = dZ·W2  [M,F]  and then  dY[:, :F] = dout·gelu(gate),  dY[:, F:] = dout·hidden·gelu'(gate).
 * geglu_linear_bwd does both in ONE launch: dout never goes to memory (it is rounded to `dtype` in the tile, exactly as the
 * separate tensor would be) and dY [M,2F] is what lora_linear_bwd_input / lora_grad_batched of `proj` consume.
 * W2t = W2ᵀ [F, Nz] row-major in `dtype` (lora_cast_matrix, transpose); zeros: at least 32·max(Nz, F) zero bytes (the launch
 * has no rank-r term).  16-bit dtypes, Nz % 64 == 0, F % 128 == 0; otherwise LORA_E_UNSUPPORTED (caller: a GEMM of its own
 * followed by geglu_gate_bwd).
 */
int geglu_linear_bwd(const void* dZ, const void* W2t, const void* Y, void* dY, const void* zeros, int64_t M, int Nz,
                     int F, int dtype, void* stream);

/*
 * Backward w.r.t. the input — autograd of lora.py:49-50 as driven by
 * training_scripts/train_lora_dreambooth.py:877 (accelerator.backward), base W frozen (:595):
 *     U  = dY·B                         [M,r]  (written to U_out, fp32, unscaled)
 *     dX = dY·W + s·U·A                 [M,K]  (skipped when dX == NULL: attn2 to_k/to_v with a
 *                                               frozen text encoder need no input gradient)
 * Wt is the frozen weight stored TRANSPOSED, Wt[K,N] = Wᵀ, so that the contraction index n is
 * contiguous for both operands (the caller caches Wt once per frozen layer).  Apack, Bpack as above.
 */
int lora_linear_bwd_input(const void* dY, const void* Wt, const float* A, const float* B,
                          const void* Apack, const void* Bpack, void* dX /* nullable */, float* U_out,
                          int64_t M, int K, int N, int r, float scale, int dtype, void* stream);
/*
 * The same with caller-provided scratch for split-K: a contraction on a grid too small for the chip (the GEGLU `proj`
 * backward: dX[1024,1280] = dY[1024,10240]·W — 80 output tiles, 160 K-steps; the 1280-wide projections at 1024 and 256
 * rows) is cut into K-slices inside ONE launch: a slice stores its fp32 partial tile, takes a ticket of the tile, and the
 * workgroup that draws the last ticket adds the slices in index order (deterministic whoever arrives last) and runs the
 * usual epilogue (rank-r term, bias, store).
 * lora_gemm_workspace_bytes(M, Kc, Nc, dtype) says how much scratch the contraction [M,Kc]·[Nc,Kc]ᵀ wants (0: the
 * library would not split it).  Layout: the first LORA_GEMM_WS_TICKET_BYTES bytes are the tiles' tickets — they must be
 * ZERO when the call is made and are zero again when its launch has finished (so one buffer, zeroed once, serves every
 * later call on the same stream) — the rest is uninitialised scratch.  16-byte aligned; used only by the call's launch
 * (stream-ordered); two launches that may run concurrently need a workspace each.
 */
#define LORA_GEMM_WS_TICKET_BYTES 4096
int64_t lora_gemm_workspace_bytes(int64_t M, int Kc, int Nc, int dtype);
int lora_linear_bwd_input_ws(const void* dY, const void* Wt, const float* A, const float* B,
                             const void* Apack, const void* Bpack, void* dX /* nullable */, float* U_out,
                             int64_t M, int K, int N, int r, float scale, int dtype, void* workspace,
                             int64_t ws_bytes, void* stream);

/*
 * The fused kernel's own contract, for callers that hold PACKED factors only — grouped layers that share an input:
 *     C[M,Nc] = Am·Bmᵀ + bias + s·P·Qᵀ ,   P[M,r] = Am·Fᵀ                       (P stored unscaled, fp32)
 * Am [M,Kc] with row stride lda (elements; 0 = Kc), Bm [Nc,Kc], Fp [16,Kc] and Qp [Nc,16] in `dtype` with the
 * unused rank rows / columns zero.  C == NULL computes P only (backward of a layer whose input needs no gradient).
 * Uses in the product (lora_diffusion/lora.py:49-50 and its autograd, several LoraInjectedLinear at once):
 *   - attn1 to_q/to_k/to_v, one launch each way: Bm = [Wq;Wk;Wv] (forward) or its transpose (backward), rank 3r with
 *     block-diagonal factors — X is read once instead of three times, dX needs no accumulation;
 *   - the attn2 to_k/to_v of ALL transformer blocks, which multiply the same encoder_hidden_states: forward as one
 *     launch over the concatenated weights with per-part factors (tile_part[column tile of 64] = part | first<<16,
 *     Fp = [n_parts·16, Kc], P_out = [n_parts][M][r]); backward (no dX: the text encoder output is frozen) as one
 *     P-only launch with part_table[g] = {column offset into Am, contraction length, offset into Fp, offset into P_out}.
 * work_cols: Σ contraction lengths of a part_table launch (profiler accounting only; 0 = Kc).
 * workspace / ws_bytes: optional split-K scratch (lora_gemm_workspace_bytes; NULL / 0 = never split).
 * Returns LORA_E_UNSUPPORTED when the operands are not 16-byte aligned / not a multiple of one K-step (grouping is
 * then simply not used by the caller).
 */
int lora_gemm_packed(const void* Am, int64_t lda, const void* Bm /* nullable with C */, const void* bias /* nullable */,
                     const void* Fp, const void* Qp, const int* tile_part /* nullable, device */,
                     const int64_t* part_table /* nullable, device */, int n_parts, void* C /* nullable */,
                     float* P_out, int64_t M, int Kc, int Nc, int r, float scale, int64_t work_cols,
                     void* workspace /* nullable */, int64_t ws_bytes, int dtype, void* stream);

/*
 * The same contract for n_parts (2..4) LoraInjectedLinear layers of EQUAL shape that share an input, at any rank r <= 16 —
 * attn1 to_q/to_k/to_v (and CLIPAttention q_proj/k_proj/v_proj) when 3r no longer fits one 16-slot factor, i.e. the ranks of
 * BASELINE configs 3 (r = 8, train_lora_dreambooth.py:596-613) and 5 (r = 16, cli_lora_pti.py:693): three
 * lora_diffusion/lora.py:49-50 forwards, or their three backward-input products summed, in ONE launch.  f16 / bf16.
 *   parts_on_k == 0 (forward):  C[M,Nc] = Am·Bmᵀ + bias + s·P_g·Q_gᵀ on column run g = [g·Nc/n, (g+1)·Nc/n),
 *       Am = X [M,Kc], Bm = [W_0; …] [Nc,Kc], Fp = [n_parts][16,Kc] (part g: rows j < r = A_g), Qp = [Nc,16] (row of part
 *       g: columns j < r = B_g); P_out[m, g·r + j] = (X·A_gᵀ)[m, j], row stride ldp floats (>= n_parts·r).
 *   parts_on_k == 1 (backward-input, n_parts == 3):  C[M,Nc] = Am·Bmᵀ + s·Σ_g P_g·Q_gᵀ with the CONTRACTION cut into runs,
 *       Am = [dY_0 | …] [M,Kc], Bm = [W_0; …]ᵀ [Nc,Kc], Fp = [16,Kc] (column run g, rows j < r = B_gᵀ), Qp = [n_parts][Nc,16]
 *       (part g: columns j < r = A_gᵀ); P_out[m, g·r + j] = (dY_g·B_g)[m, j].  No accumulation of three dX tensors.
 * Unused rank rows / columns of Fp / Qp are zero.  LORA_E_UNSUPPORTED: f32, part width or Kc not a multiple of 64, operands
 * not 16-byte aligned, parts_on_k with n_parts != 3 (the caller then runs the layers one by one).
 */
int lora_gemm_parts(const void* Am, const void* Bm, const void* bias /* nullable */, const void* Fp, const void* Qp,
                    void* C, float* P_out, int64_t ldp, int64_t M, int Kc, int Nc, int r, int n_parts, int parts_on_k,
                    float scale, int dtype, void* stream);

/*
 * Backward w.r.t. the LoRA factors (no grad for W or b: lora.py:179-180 set requires_grad only on
 * lora_up / lora_down; train_lora_dreambooth.py:595 freezes the rest):
 *     gB = s·dYᵀ·T      [N,r]
 *     gA = s·Uᵀ·X       [r,K]
 * The M rows are cut into `n_blocks` row blocks; block b STORES its partial sums at
 * gA_part + b·part_stride and gB_part + b·part_stride (fp32; every block is written, zeros included).
 * No global atomics: the outputs are a few KB wide and hundreds of workgroups adding onto so few cache
 * lines serialise at the memory side.  lora_reduce_partials then sums the blocks in index order
 * (deterministic): grads[i] (+)= Σ_b partials[b·part_stride + i] for i < n.  A trainer lays the partials
 * of all layers out as [n_blocks][slab] and reduces the whole slab — the RCCL all-reduce buffer — in
 * one launch per step.
 */
int lora_linear_bwd_params(const void* dY, const void* X, const float* T, const float* U,
                           float* gA_part, float* gB_part, int64_t part_stride, int n_blocks,
                           int64_t M, int K, int N, int r, float scale, int dtype, void* stream);
int lora_reduce_partials(const float* partials, int64_t part_stride, int n_blocks, float* grads,
                         int64_t n, int accumulate, void* stream);

/*
 * The same two reductions for MANY layers at once — what a training step calls once, after backward, with all
 * 2×144 problems of an SD1.5 UNet (MI355X-first: 288 GB of HBM keep every layer's dY and X alive until then, and a
 * few chip-filling launches replace 144 latency-bound ones).  One problem is  G[c,j] = s·Σ_m S[m,c]·P[m,j]:
 *     S  : streamed operand [M, C] (dtype), row stride s_stride elements — dY for gB, X for gA; may be a strided
 *          slice of a wider buffer (grouped projections);
 *     P  : [M, r] fp32, row stride p_stride floats — T for gB, U for gA;
 *     out: partial output of row block 0; rank columns j are split into groups of `rg`
 *          (out[j / rg], local column j % rg): one group normally (rg = r); a grouped q/k/v layer hands U as
 *          [M, 3r] and receives three gA outputs;  out_kn = 1 → out[g] is [rg, C] (gA), 0 → [C, rg] (gB);
 *     row block b of the problem stores at out[g] + b·part_stride; n_blocks = 0 lets the library choose
 *          (lora_grad_row_blocks(M), at most LORA_GRAD_MAX_BLOCKS) — the caller folds that many.
 * `problems` is HOST memory and is consumed during the call (the tables travel as kernel arguments, ≤ 28 problems
 * per launch; nothing is uploaded, so the call may be recorded into a hipGraph).  lora_fold_partials then sums the
 * row blocks of a table of slab ranges, each with its own block count:
 *     grads[off+i] (+)= Σ_{b < blocks} partials[b·part_stride + off + i],  ranges[k] = {off, len, blocks, 0}
 * (int64, DEVICE memory), max_len = the largest len.  Deterministic like lora_reduce_partials.
 * Arithmetic: fp32 sums.  16-bit operands are multiplied on the matrix cores with the row index as the contraction; P
 * enters as an exact hi + lo pair of 16-bit values (fp16: 22 significant bits of P, bf16: 16).  fp32 operands: VALU FMAs.
 */
#define LORA_GRAD_MAX_BLOCKS 64
typedef struct lora_grad_problem {
    const void* S;
    const float* P;
    float* out[4];
    int64_t s_stride, p_stride, part_stride, M;
    int C, r, rg, out_kn, n_blocks;
    float scale;
} lora_grad_problem;
int lora_grad_row_blocks(int64_t M);
int lora_grad_batched(const lora_grad_problem* problems, int n, int dtype, void* stream);
/*
 * The same problems in ONE launch (16-bit operands, ranks <= 16): lora_grad_batched's launches of <= 28 problems each end on a
 * tail, and the last few of a step hold a few dozen workgroups.  Here the table is a PLAN in device memory:
 *     lora_grad_plan_bytes(problems, n)  upper bound of the plan's size in bytes;
 *     lora_grad_plan(...)                writes the plan — [items: 128 bytes per problem | one int per workgroup] — into HOST
 *                                        memory `plan_host` and returns the counts; the first
 *                                        n_items·128 + n_blocks·4 bytes are what the device needs.  LORA_E_UNSUPPORTED (fp32
 *                                        operands, an unaligned operand, a rank above 16): use lora_grad_batched;
 *     lora_grad_planned(plan_dev, ...)   launches on a DEVICE copy of those bytes.  The caller makes that copy with its own
 *                                        stream-ordered transfer (inside a recording it is a memcpy node whose host source the
 *                                        caller keeps alive and unchanged for as long as the recording is replayed) — the
 *                                        library still allocates and retains nothing.  bytes / flops: what lora_prof_* charge.
 * Same arithmetic, same row-block partial layout, same fold as lora_grad_batched: bit-identical results.
 */
int64_t lora_grad_plan_bytes(const lora_grad_problem* problems, int n);
int lora_grad_plan(const lora_grad_problem* problems, int n, int dtype, void* plan_host, int64_t plan_bytes, int* n_items,
                   int* n_blocks);
int lora_grad_planned(const void* plan_dev, int n_items, int n_blocks, int dtype, double bytes, double flops, void* stream);
int lora_fold_partials(const int64_t* ranges, int n_ranges, int64_t max_len, const float* partials,
                       int64_t part_stride, float* grads, int accumulate, void* stream);

/*
 * DDPM noise-prediction loss, forward + gradient in one pass —
 * training_scripts/train_lora_dreambooth.py:855-875 and lora_diffusion/cli_lora_pti.py:222-247.
 *   rows [0, n_inst)            instance part : mean_b(mean_chw((p-t)²))
 *   rows [n_inst, n_inst+n_prior) prior part  : prior_weight · mean((p-t)²)     (n_prior may be 0)
 *   mask (nullable, fp32 [rows, 1, H, W] broadcast over `channels`, already normalised by
 *   lora_mask_prepare): p and t are multiplied by it first (cli_lora_pti.py:243-247).
 * loss_out[0] = loss (fp32).  dpred (nullable, same dtype as pred) = grad_scale · dloss/dpred.
 * workspace: at least lora_mse_workspace_bytes() bytes, 16-byte aligned; the call zeroes what it uses.
 */
int ddpm_mse_fwd_bwd(const void* pred, const void* target, const float* mask /* nullable */,
                     int n_inst, int n_prior, int64_t per_row /* C·H·W */, int64_t hw /* H·W */,
                     float prior_weight, float grad_scale, float* loss_out, void* dpred /* nullable */,
                     void* workspace, int dtype, void* stream);
int64_t lora_mse_workspace_bytes(void);

/*
 * Mask preparation of cli_lora_pti.py:222-241:
 *   out = nearest_resize(mask[B,1,Hin,Win] -> [B,1,H,W]) + 0.05 ;  out /= mean(out)
 * mask_in / mask_out fp32.  Single launch, deterministic.
 */
int lora_mask_prepare(const float* mask_in, float* mask_out, int B, int Hin, int Win, int H, int W,
                      void* stream);

/*
 * weight_apply_lora — lora_diffusion/lora.py:410-424:   W ← W + α·(B @ A).type(W.dtype)
 * In place on W[N,K] (dtype); A[r,K], B[N,r] passed as fp32.  The reference forms B @ A in the dtype the
 * factors are held in (fp16 when they come from a `.pt` file): `factor_dtype` names it, and the product
 * is rounded to it before the cast to W's dtype, the multiply by α and the add — op-by-op as the reference.
 */
int lora_merge_weight(void* W, const float* A, const float* B, int K, int N, int r, float alpha,
                      int dtype, int factor_dtype, void* stream);

/* The same merge for every layer of a model in ONE launch (cli_lora_add.py:72-88 `upl`: 192.6 M base weights of an
 * SD1.5 UNet, one HBM pass).  table: device memory, int64 [n_layers][8] =
 * {W ptr, A ptr (fp32 [r,K]), B ptr (fp32 [N,r]), K, N, r, dtype of W, factor_dtype}; max_elems = max N·K. */
int lora_merge_weight_batched(const int64_t* table, int n_layers, int64_t max_elems, float alpha, void* stream);

/*
 * LoRA (+) LoRA interpolation — lora_diffusion/cli_lora_add.py:52-55 (mode `lpl`), op by op in the tensors' dtype:
 *     x1 ← T( T(a·x1) + T(b·x2) )        with a = alpha, b = 1 − alpha
 * over n elements (the caller concatenates a whole `[up0, down0, …]` list into one buffer).
 */
int lora_lerp(void* x1, const void* x2, int64_t n, float a, float b, int dtype, void* stream);

/* Out-of-place transpose+cast helper used to build the cached operands:
 * dst[cols,rows] (dst_dtype) = src[rows,cols] (src_dtype)ᵀ ; transpose=0 gives a plain cast. */
int lora_cast_matrix(const void* src, void* dst, int64_t rows, int64_t cols, int src_dtype,
                     int dst_dtype, int transpose, void* stream);

/*
 * Fused clip_grad_norm_ + AdamW over the flat LoRA slab —
 * training_scripts/train_lora_dreambooth.py:878-888 (clip_grad_norm_(…, max_grad_norm); optimizer.step())
 * and lora_diffusion/cli_lora_pti.py:448-451.
 *   lora_grad_sqnorm : norm_out (4 floats, device; zero it once before the first step):
 *                      [0] = Σ (grad_mul·g)² over n elements (deterministic two-level sum),
 *                      [1] = 1.0f if any element is non-finite else 0.0f,
 *                      [2] += 1 when [1] == 0: the number of APPLIED optimizer steps including this one,
 *                      [3] += 1 when [1] != 0: the number of skipped steps.
 *   lora_adamw_step  : g' = grad_mul·g·min(1, max_norm/(sqrt(norm_in[0])+1e-6))   (max_norm<=0: no clip)
 *                      torch.optim.AdamW update (decoupled weight decay, bias correction with `step`, or with
 *                      the device counter norm_in[2] when step == 0), skipped entirely when norm_in[1] != 0 —
 *                      torch.cuda.amp.GradScaler semantics: an overflowed step neither moves the parameters nor
 *                      advances the optimizer's step count.
 * `grad_mul` carries 1/world_size (mean all-reduce) and 1/loss_scale.
 */
int lora_grad_sqnorm(const float* grad, int64_t n, float grad_mul, float* norm_out, void* workspace,
                     void* stream);
int64_t lora_sqnorm_workspace_bytes(void);
int lora_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    const float* norm_in /* nullable: no clip, no skip */, float grad_mul, float max_norm,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    void* stream);

/*
 * Step prologue — training_scripts/train_lora_dreambooth.py:824-853 (add_noise / target selection)
 * with the DDPM definitions (diffusers DDPMScheduler, not vendored in the reference):
 *     noisy  = sqrt_acp[t_b]·x0 + sqrt_1macp[t_b]·eps
 *     target = eps                                   (v_prediction == 0)
 *            = sqrt_acp[t_b]·eps − sqrt_1macp[t_b]·x0 (v_prediction != 0)
 * x0, eps fp32 [B, per_row]; t int64 [B]; tables fp32 [T]; noisy/target in `dtype`.
 */
int ddpm_add_noise(const float* x0, const float* eps, const int64_t* t, const float* sqrt_acp,
                   const float* sqrt_1macp, void* noisy, void* target, int B, int64_t per_row,
                   int v_prediction, int dtype, void* stream);

/*
 * Step prologue with build-owned, counter-based randomness: replaces randn_like + randint +
 * add_noise (+ get_velocity) of training_scripts/train_lora_dreambooth.py:824-853 by one launch.
 *     t_b ~ U{0..n_timesteps-1},  eps ~ N(0,1)   from Philox4x32-10 keyed by (seed, step): the same draw on every
 *     rank (set_seed semantics, :509-510) and on the CPU oracle (oracle/philox.py);
 *     noisy / target as in ddpm_add_noise.  eps_out (fp32) and t_out (int64) are optional copies of the draw.
 * Counter layout: element group g = i/4 uses counter (g, 0, 0, 0) → 4 normals by Box–Muller; row b uses
 * counter (b, 0, 1, 0), word 0, mapped to [0, n_timesteps) by the high half of a 32×32-bit product.
 */
int ddpm_noise_prologue(const float* x0, const float* sqrt_acp, const float* sqrt_1macp, void* noisy,
                        void* target /* nullable */, float* eps_out /* nullable */, int64_t* t_out /* nullable */,
                        int B, int64_t per_row, int n_timesteps, uint64_t seed, uint64_t step,
                        int v_prediction, int dtype, void* stream);

/*
 * The two ops sandwiched by the hot path inside a transformer block, as streaming kernels.
 *   geglu_gate_fwd : out[M,C]  = h · gelu(g)  with [h | g] = y[M,2C], exact (erf) gelu — the body of diffusers'
 *                    GEGLU.forward, the caller of the `proj` LoraInjectedLinear (target class "GEGLU", lora.py:53).
 *   geglu_gate_bwd : dy[M,2C] = [dout·gelu(g) | dout·h·gelu'(g)], contiguous, consumed directly as dY by
 *                    lora_linear_bwd_input / lora_linear_bwd_params of `proj`.
 *   attn_split_heads : [B,N,H·d] → [B,H,N,D], D >= d, columns d..D-1 zero-filled (q/k/v into the attention core).
 *   attn_merge_heads : [B,H,N,D] → [B,N,H·d] (the core's output back, padding dropped).  Each is the other's
 *                    backward.  d and D multiples of 16 bytes.
 */
int geglu_gate_fwd(const void* y, void* out, int64_t M, int C, int dtype, void* stream);
int geglu_gate_bwd(const void* y, const void* dout, void* dy, int64_t M, int C, int dtype, void* stream);
int attn_split_heads(const void* src, void* dst, int B, int N, int H, int d, int D, int dtype, void* stream);
int attn_merge_heads(const void* src, void* dst, int B, int N, int H, int d, int D, int dtype, void* stream);
/* same, reading a [B,H,N,D] VIEW whose (b,h,n) rows start at b·sB + h·sH + n·sN elements (last dim contiguous):
 * the attention core hands back its output and its q/k/v gradients as transposed views; this consumes them in place. */
int attn_merge_heads_strided(const void* src, void* dst, int B, int N, int H, int d, int D, int64_t sB, int64_t sH,
                             int64_t sN, int dtype, void* stream);

/*
 * Token-embedding rows for a text encoder whose INPUT EMBEDDINGS train next to the UNet's LoRA factors: the tuning phase of
 * lora_diffusion/cli_lora_pti.py with continue_inversion (default, :528) puts `text_encoder.get_input_embeddings().parameters()`
 * in the optimizer (:706-722) and runs `text_encoder(batch["input_ids"])[0]` inside loss_step (:199-206) — BASELINE config 5,
 * "+ extended-latent TI".  They replace `torch.nn.Embedding.forward` / its backward for that one table (fp32 master [V, D]).
 *   embed_rows_fwd : out[p, :] = table[ids[p], :] cast to out_dtype, p < n (ids int64; out of range ids are clamped).
 *   embed_rows_bwd : grad_table[t, :] (+)= Σ_{p : ids[p] == t} dE[p, :] for every token t that occurs, the positions added in
 *                    ascending order by ONE owner workgroup (torch's scatter uses atomics: run-to-run sum order).  Rows of tokens
 *                    that do not occur are not touched.  dE in `dtype`; accumulate = 0 overwrites the rows that occur; active[t] = 1
 *                    for every token that occurs (when given).
 */
int embed_rows_fwd(const float* table, const int64_t* ids, void* out, int64_t n, int D, int64_t V, int out_dtype, void* stream);
int embed_rows_bwd(const void* dE, const int64_t* ids, float* grad_table, unsigned char* active /* nullable: [V] */, int64_t n,
                   int D, int64_t V, int dtype, int accumulate, void* stream);
/* torch.optim.AdamW (`lora_adamw_step`'s arithmetic, same norm / skip / step-count inputs) over a [V, D] table of which only the
 * rows flagged in `active` (set by embed_rows_bwd, never cleared) have ever had a gradient: those get the full update, every
 * other row p ← p·(1 − lr·wd) — bit-identical to the dense update (g = m = v = 0 there), at a third of its traffic.
 * Reference: the token table as an AdamW group, cli_lora_pti.py:706-738, stepped at :451. */
int lora_adamw_rows(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const unsigned char* active, int64_t V,
                    int D, const float* norm_in, float grad_mul, float max_norm, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, void* stream);

/*
 * Short-context attention core  O = softmax(Q·Kᵀ·scale)·V  per head, for at most 128 keys: the cross-attention
 * (`attn2`) between the to_q/to_k/to_v and to_out LoRA linears (diffusers CrossAttention.forward, the
 * caller of the layers wrapped by lora_diffusion/lora.py:137-183).  Tensors keep the layout those linears produce
 * and consume — Q/O/dO/dQ [B, Tq, H·d], K/V/dK/dV [B, Tk, H·d] — so there are no head split/merge copies.
 *   attn_ctx_supported          : 1 when (shape, dtype) runs here: f16/bf16, d % 8 == 0, d <= 160, Tk <= 128 (<= 96 when d > 96).
 *   attn_ctx_fwd                : O.  Nothing else is saved: backward recomputes the single key tile.
 *   attn_ctx_bwd_workspace_bytes: size of the fp32 partial-sum workspace for dK/dV (-1 when unsupported).
 *   attn_ctx_bwd                : dQ, dK, dV from Q, K, V, dO.  Deterministic (ordered partial sums, no atomics).
 * Unsupported shapes return LORA_E_BADARG; the caller keeps its generic attention for those.
 */
int attn_ctx_supported(int B, int Tq, int Tk, int H, int d, int dtype);
/* _strided forms: K and V (resp. dK and dV) are [B·Tk, ·] column slices of a wider row-major buffer with row stride
 * ldk (ld_dk) elements — the output (gradient) buffer of a grouped to_k/to_v projection (lora_gemm_packed). */
int attn_ctx_fwd_strided(const void* Q, const void* K, const void* V, void* O, int64_t ldk, int B, int Tq, int Tk,
                         int H, int d, float scale, int dtype, void* stream);
int attn_ctx_bwd_strided(const void* Q, const void* K, const void* V, const void* dO, void* dQ, void* dK, void* dV,
                         void* workspace, int64_t ldk, int64_t ld_dk, int B, int Tq, int Tk, int H, int d,
                         float scale, int dtype, void* stream);
int attn_ctx_fwd(const void* Q, const void* K, const void* V, void* O, int B, int Tq, int Tk, int H, int d,
                 float scale, int dtype, void* stream);
int64_t attn_ctx_bwd_workspace_bytes(int B, int Tq, int Tk, int H, int d);
int attn_ctx_bwd(const void* Q, const void* K, const void* V, const void* dO, void* dQ, void* dK, void* dV,
                 void* workspace, int B, int Tq, int Tk, int H, int d, float scale, int dtype, void* stream);

/*
 * Long-context attention core (self-attention: thousands of keys), same tensor layouts as attn_ctx_*: flash-style
 * online softmax over 64-key tiles, no [Tq, Tk] matrix in memory.
 *   attn_flash_supported          : 1 for f16/bf16, d % 8 == 0, d <= 160.
 *   attn_flash_fwd                : O and LSE [B, H, Tq] fp32 = log2-sum-exp2 of the scaled scores (saved for backward;
 *                                   may be NULL when no backward follows).
 *   attn_flash_bwd_workspace_bytes: B·H·Tq·4 (the softmax correction Δ = Σ dO·O per query row).
 *   attn_flash_bwd                : dQ, dK, dV from Q, K, V, O, dO, LSE.  Two launches (query-owned dQ, which also fills the
 *                                   workspace with Δ; key-owned dK/dV), every output element written by one workgroup:
 *                                   deterministic, no atomics.
 */
int attn_flash_supported(int B, int Tq, int Tk, int H, int d, int dtype);
/* _strided forms: Q, K, V share the row stride ldq and dQ, dK, dV the row stride ld_dq (elements): the three column
 * slices of a grouped to_q/to_k/to_v projection's output / gradient buffer.  O, dO, LSE stay dense. */
int attn_flash_fwd_strided(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t ldq, int B,
                           int Tq, int Tk, int H, int d, float scale, int dtype, void* stream);
int attn_flash_bwd_strided(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                           const float* LSE, void* dQ, void* dK, void* dV, void* workspace, int64_t ldq,
                           int64_t ld_dq, int B, int Tq, int Tk, int H, int d, float scale, int dtype, void* stream);
int attn_flash_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int Tq, int Tk, int H,
                   int d, float scale, int dtype, void* stream);
int64_t attn_flash_bwd_workspace_bytes(int B, int Tq, int H);
int attn_flash_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                   void* dQ, void* dK, void* dV, void* workspace, int B, int Tq, int Tk, int H, int d, float scale,
                   int dtype, void* stream);

/*
 * Launch profiler (measurement only; off by default).  When enabled, the hot-path kernels are launched
 * with start/stop events attached to the dispatch itself, so each record is that kernel's own duration on
 * the caller's stream, together with the ALGORITHMIC bytes and flops of the call (formulas: DESIGN.md §5).
 * Records are grouped per kernel instantiation (the names rocprofv3 shows); lora_prof_kernel_name(i)
 * returns a substring of that name.  lora_prof_collect waits for the recorded events, returns the totals
 * and resets the recording.
 */
#define LORA_PROF_KINDS 18
typedef struct lora_prof_totals {
    int64_t launches[LORA_PROF_KINDS];
    double ms[LORA_PROF_KINDS];
    double bytes[LORA_PROF_KINDS];
    double flops[LORA_PROF_KINDS];
} lora_prof_totals;
int lora_prof_enable(int capacity /* max recorded launches; 0 disables and frees */);
int lora_prof_collect(lora_prof_totals* out);
const char* lora_prof_kernel_name(int kind);
/*
 * Launch-floor mode (measurement only; bench.py's `roofline.step_level.launch_floor_ms`): while on, every profiled launch site
 * dispatches an EMPTY kernel of the same signature — same grid, block, dynamic LDS and kernel-argument segment; it reads two
 * argument words and returns — in place of the real one.  With the profiler enabled the recorded durations are then what a
 * launch of that shape costs before it does any work.  Outputs are NOT written while the mode is on: timing only.
 */
int lora_prof_null_mode(int on);

#ifdef __cplusplus
}
#endif
#endif /* LORA_HIP_H */
