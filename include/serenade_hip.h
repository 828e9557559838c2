/*
 * serenade_hip.h — C ABI of libserenade_hip.so (MI355X / gfx950 only).
 *
 * The reference (imulki/serenade) is pure Python on PyTorch and has NO FFI or operator
 * registry: its boundary is the Python class API.  This header is
 * therefore the boundary this build defines *behind* that API: one entry point per
 * arithmetic stage of the inference hot path, each citing the reference code it replaces.
 * The Python mirror classes (serenade_amd.models / serenade_amd.vocoder) keep the
 * reference's signatures and state_dict layout and call only these functions.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (fp32 unless noted), borrowed for the call only;
 *   - activations are channels-last: (batch, time, channels), channels contiguous;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, negative on error; srn_last_error() gives the message
 *     (thread-local).  No allocation, no global mutable state, re-entrant.
 */
#ifndef SERENADE_HIP_H
#define SERENADE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRN_ABI_VERSION 3
#define SRN_MAX_TAPS 16

/* prologue activation applied to the gathered input elements */
enum { SRN_ACT_NONE = 0, SRN_ACT_LEAKY = 1, SRN_ACT_SILU = 2, SRN_ACT_MISH = 3 };
/* residual mode */
enum { SRN_RES_NONE = 0, SRN_RES_ADD = 1, SRN_RES_AXPY = 2 };
/* post op */
enum {
  SRN_POST_NONE = 0,
  SRN_POST_DIV = 1, /* v / post_div (HiFi-GAN: cs / num_blocks, hifigan.py:186) */
  SRN_POST_TANH = 2,
  SRN_POST_RELU = 3,
  SRN_POST_LEAKY = 4 /* LeakyReLU with slope post_div */
};
/* arithmetic of the contraction */
enum {
  SRN_PREC_FP32 = 0,  /* exact fp32 MFMA */
  SRN_PREC_BF16X3 = 1, /* split-bf16 (hi + lo), 3 MFMA per product, fp32 accumulate: ~2^-17 per product */
  SRN_PREC_BF16X6 = 2  /* fp32-faithful emulation: exact (hi + mid + lo) split, 6 MFMA per product, fp32 accumulate:
                        * <= 2^-26 per product (below fp32's own rounding unit).  Kernels without a bf16x6 variant
                        * (generic conv_gemm, halo / strip, srn_hifigan_resunit) run their exact-fp32 path instead. */
};

/*
 * Generalised implicit-GEMM "conv1d" on channels-last fp32 tensors.  Arithmetic of the contraction per
 * `precision`: SRN_PREC_FP32 = exact fp32 on v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fma chain, the
 * reference's precision); SRN_PREC_BF16X3 = every fp32 operand split into (hi, lo) bf16, three
 * v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate (~2^-17 relative per product):
 *
 *   out[z, t*out_t_stride + out_t_off, n] = epilogue( alpha * sum_{tap, c} w[n, tap*C_in + c] *
 *                                  act( in[z, t*in_stride + tap_off[tap], c] ) + bias[n] )
 *
 * Covers, with different parameters, every contraction on the hot path:
 *   Conv1d k3/k7/k11 (+dilation, stride 2, zero or reflect padding)   decoder.py:70,107,271,335;
 *                                                                      serenade.py:282-294,366-373;
 *                                                                      hifigan.py:71-77,140-148; residual_block.py:187-226
 *   ConvTranspose1d (one call per output phase)                        decoder.py:191; hifigan.py:96-104
 *   Conv1d k1 / nn.Linear                                              decoder.py:90,341; transformer.py:120-138
 *   QK^T and PV of the self-attention (batched over batch x heads)     transformer.py:292-301 (diffusers Attention)
 * Input rows t >= len_in[zb] are read as zero (the reference's `x * mask`), rows outside
 * [0, T_in) are zero or reflected.  The input may be the channel-concatenation of two tensors
 * (decoder.py:405,449 `pack([x, skip])`) without materialising it.
 */
typedef struct SrnConvParams {
  int32_t n_batch;   /* z range = n_batch * n_head */
  int32_t n_head;    /* >= 1; z -> (zb = z / n_head, zh = z % n_head) */
  int32_t T_in, T_out;
  int32_t C_in;      /* channels per tap seen by the input gather (multiple of 4) */
  int32_t C_in0;     /* channels taken from in0; the remaining C_in - C_in0 come from in1 (multiple of 32 if < C_in) */
  int32_t C_w;       /* valid k per tap on the weight side (<= C_in; = C_in normally) */
  int32_t N;         /* GEMM N (weight rows) */
  int32_t N_out;     /* stored output channels (N, or N/2 for GEGLU) */
  int32_t n_taps;
  int32_t tap_off[SRN_MAX_TAPS];
  int32_t in_stride;
  int32_t pad_reflect; /* 0: zero padding, 1: reflection at the tensor's ends, 2: reflection at 0 and at the item's own
                        * end len_in[zb] (exact ragged batches: what nn.ReflectionPad1d does to the unpadded item) */
  int32_t w_nmajor;    /* 0: w is [N][n_taps*C_in] (k contiguous); 1: w is [K][N] (n contiguous), n_taps == 1 */
  int32_t pro_act;
  float pro_slope;
  float alpha;
  float beta;        /* SRN_RES_AXPY: out = res + beta * val */
  int32_t geglu;     /* 1: weight rows interleaved in 32-row (value | gate) groups, out = value * gelu(gate) */
  int32_t res_mode;
  int32_t post;
  float post_div;
  int32_t out_t_stride, out_t_off;
  int32_t tile;      /* 0 = auto, else a tile-config id (see conv_gemm.hip) */
  const float* in0; int64_t in0_bs, in0_hs; int32_t ld_in0;
  const float* in1; int64_t in1_bs; int32_t ld_in1;
  const float* w;   int64_t w_bs, w_hs; int32_t ldw;
  const float* bias;
  const int32_t* len_in;   /* per zb, or NULL */
  const int32_t* len_out;  /* per zb, or NULL: output rows t >= len_out are multiplied by 0 */
  const float* res;  int64_t res_bs, res_hs; int32_t ld_res;
  const float* res2; int64_t res2_bs; int32_t ld_res2;  /* second additive residual (HiFi-GAN stage sum) */
  float* out; int64_t out_bs, out_hs; int32_t ld_out;
  int32_t precision;  /* SRN_PREC_* */
  int32_t no_halo;    /* kernel selection behind this entry point (testing / A-B timing): 0 = automatic, 1 = tiled kernels
                       * only (no halo, no strip), 2 = halo kernel whenever eligible, 3 = generic conv_gemm kernel only,
                       * 4 = strip kernel whenever eligible, 5 = exact fp32 on conv_fast.hip's loop instead of conv_f32.hip's */
  /* ws / ws_bytes: optional caller-owned workspace (16-byte aligned, at least srn_conv_gemm_workspace_bytes(p)
   * bytes, shared by all calls of one stream).  With it, launches whose tile grid cannot fill the chip are split
   * over K (conv_splitk.hip); without it (NULL) they run unsplit.  Results agree to fp32 summation order.
   * w_lo (SRN_PREC_BF16X6 only): the lo plane of the pre-split weights, bf16 [N][n_taps][roundup(C_in, 32) / 32][32];
   *   w_hi then holds [hi 32 | mid 32].  NULL otherwise.
   * w_hi (SRN_PREC_BF16X3 only): the static weights already split at load time, bf16
   * [N][n_taps][roundup(C_in, 32) / 32][hi 32 | lo 32]; NULL: the kernel splits the fp32 rows of `w` per call. */
  void* ws; int64_t ws_bytes;
  const void* w_hi; const void* w_lo;
  float* gn_partials; /* or NULL: [zb][ceil(T_out/32)][N/32][2] per-32x32-tile (sum, sumsq) of the stored values */
  /* Transposed tail, or out_tr == NULL: GEMM columns c >= out_tr_col0 (a multiple of 32) are written to
   * out_tr[zb][c - out_tr_col0][t] (row stride ld_out_tr, batch stride out_tr_bs) INSTEAD of `out`.  The QKV
   * projection writes V^T with it, so P.V (transformer.py:292-301) contracts k-major rows like every other GEMM and
   * no transpose pass runs.  Plain epilogue only (no GEGLU / residual / post op / row stride / heads). */
  float* out_tr; int64_t out_tr_bs; int32_t ld_out_tr; int32_t out_tr_col0;
} SrnConvParams;

int srn_abi_version(void);
const char* srn_last_error(void);

/* the workhorse above */
int srn_conv_gemm(const SrnConvParams* p, void* stream);
/* bytes of workspace the split-K path would use for this call (0: the call is never split) */
int64_t srn_conv_gemm_workspace_bytes(const SrnConvParams* p);

/*
 * GroupNorm(8 groups, eps) -> Mish -> (+ time_bias[c]) -> * mask   (Block1D tail + the time-embedding add of
 * ResnetBlock1D; decoder.py:71-77,96-97).  Statistics come from the per-tile partials written by
 * srn_conv_gemm and run over the full padded length T (as the reference's GroupNorm does).
 *   x, y: (B, T, C); gamma, beta: (C); time_bias: (C) at time_bias + b * time_bias_bs, or NULL; lens: (B) or NULL.
 * valid_stats = 1 (exact ragged batches): the statistics run over the item's own lens[b] rows -- the producing conv
 * must have zeroed its padded rows (len_out) so that they add nothing to the partial sums -- which makes every item of a
 * padded batch equal to its B = 1 run; 0 = the reference's batched semantics above.
 */
int srn_gn_mish_apply(const float* x, const float* gn_partials, const float* gamma, const float* beta,
                      const float* time_bias, int64_t time_bias_bs, const int32_t* lens, float* y, int B, int T,
                      int C, int groups, float eps, int valid_stats, void* stream);

/*
 * Tail of ResnetBlock1D (decoder.py:98-101, 34-45):
 *   v = Mish(GroupNorm(c2)) * mask + r ;  y = (v - mean_c) / sqrt(var_c + eps) * scale[b] + shift[b]
 * c2, r, y: (B, T, C); scale, shift: row b at scale + b * ld_ss (C values each).
 */
int srn_resblock_tail(const float* c2, const float* gn_partials, const float* gamma, const float* beta,
                      const int32_t* lens, const float* r, const float* scale, const float* shift, int64_t ld_ss,
                      float* y, int B, int T, int C, int groups, float gn_eps, float ln_eps, int valid_stats,
                      void* stream);

/* The same, and in the same launch the LayerNorm that opens the transformer block behind every ResnetBlock1D
 * (decoder.py:411-421 -> transformer.py:286, norm1):  y2 = LayerNorm(y; ln2_gamma, ln2_beta, ln2_eps), from the row still
 * in registers (one HBM pass and one launch fewer per block; y2 equals srn_layernorm(y) bit for bit). */
int srn_resblock_tail_ln(const float* c2, const float* gn_partials, const float* gamma, const float* beta,
                         const int32_t* lens, const float* r, const float* scale, const float* shift, int64_t ld_ss,
                         float* y, int B, int T, int C, int groups, float gn_eps, float ln_eps, int valid_stats,
                         const float* ln2_gamma, const float* ln2_beta, float* y2, float ln2_eps, void* stream);

/* nn.LayerNorm over the last dim (transformer.py:211,249): x, y (rows, C). */
int srn_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps,
                  void* stream);

/*
 * Row softmax of attention scores with a key-padding mask (F.scaled_dot_product_attention semantics at
 * transformer.py:292-301): s (Z, L, ld) in place; keys k >= lens[z / n_head] get probability 0;
 * columns [L, ld) are written as 0.
 */
int srn_softmax_rows(float* s, const int32_t* lens, int Z, int n_head, int L, int ld, void* stream);

/* SinusoidalPosEmb (decoder.py:54-63): out row i (stride ld) = [sin(scale*t_i*f_k), cos(scale*t_i*f_k)], t (n) on device. */
int srn_sinusoidal_emb(const float* t, float* out, int n, int dim, int ld, float scale, void* stream);

/* Generic channels-last copy with channel offset/strides: dst[b,t,dc0+c] = src[b,t,sc0+c] * a + bvec?  (pack/concat) */
int srn_copy_channels(const float* src, int64_t src_bs, int ld_src, int sc0, float* dst, int64_t dst_bs, int ld_dst,
                      int dc0, int B, int T, int C, void* stream);

/* Ragged time-concat (serenade.py:202 `cat([ref_mu, src_mu], dim=1)` per item of a batch whose prompts differ in
 * length): rows [0, n_rows[b]) of src item b -> dst rows row_off[b] + r, channels [dc0, dc0 + C).  NULL row_off = 0,
 * NULL n_rows = T. */
int srn_scatter_rows(const float* src, int64_t src_bs, int ld_src, float* dst, int64_t dst_bs, int ld_dst, int dc0,
                     const int32_t* row_off, const int32_t* n_rows, int B, int T, int C, void* stream);

/* (B, C, T) <-> (B, T, C) transposes at the API edge (reference tensors are (B, C, T); decoder.py:405-467). */
int srn_transpose_ct(const float* src, float* dst, int B, int R, int Cc, int64_t src_bs, int ld_src, int64_t dst_bs,
                     int ld_dst, void* stream);
/* Many such transposes in ONE launch (the table travels in the kernel arguments: capturable): entry e does
 * dst_e[b][c][r] = src_e[b][r][c] for b < B, r < R, c < Cc.  The training step re-lays every conv / linear weight
 * once per step (torch (N, C, k) -> the k-major rows srn_conv_gemm reads, and W^T for the input gradient; what autograd
 * does implicitly for decoder.py's Conv1d / Linear under trainers/ssc.py:57-96): ~140 launches of a few microseconds
 * each when issued one by one. */
#define SRN_TR_LIST_MAX 40
typedef struct SrnTransposeList {
  int32_t n;
  int32_t pad_;
  const float* src[SRN_TR_LIST_MAX];
  float* dst[SRN_TR_LIST_MAX];
  int64_t src_bs[SRN_TR_LIST_MAX], dst_bs[SRN_TR_LIST_MAX];
  int32_t B[SRN_TR_LIST_MAX], R[SRN_TR_LIST_MAX], Cc[SRN_TR_LIST_MAX];
  int32_t ld_src[SRN_TR_LIST_MAX], ld_dst[SRN_TR_LIST_MAX];
  int32_t first_block[SRN_TR_LIST_MAX + 1];  /* filled by the library */
} SrnTransposeList;
int srn_transpose_multi(const SrnTransposeList* list, void* stream);

/* y = (x * a[c] + b[c] - c[c]) / d[c] (Vocoder.decode normalisation, vocoder.py:52-56); x, y (rows, C). */
int srn_renorm(const float* x, const float* trg_scale, const float* trg_mean, const float* voc_mean,
               const float* voc_scale, float* y, int64_t rows, int C, void* stream);

/* HiFi-GAN output stage: LeakyReLU(slope) -> Conv1d(C -> 1, k, pad (k-1)/2) -> tanh  (hifigan.py:137-149).
 * x (B, T, C) channels-last, w (k, C), y (B, T). */
int srn_out_conv_tanh(const float* x, const float* w, const float* bias, float* y, int B, int T, int C, int k,
                      float slope, void* stream);

/*
 * One residual unit of HiFiGANResidualBlock with use_additional_convs (residual_block.py:243-258, loop body):
 *     xt = Conv1d_k,dil(LeakyReLU(x));  xt = Conv1d_k,1(LeakyReLU(xt));  y = xt + x
 * fused into ONE launch for thin stages (C = 32 or 64): the receptive-field rows of x are staged once in LDS, the
 * intermediate `xt` never leaves LDS, weights stream through a double-buffered LDS stage -- one HBM read and one
 * write per unit instead of five passes.  Optionally the stage bookkeeping of HiFiGANGenerator.forward
 * (hifigan.py:183-186) rides in the epilogue:  y = (y + res2) / post_div.
 *   x, out, res2: (n_batch, T, C) channels-last fp32, rows contiguous (ld = C); out must not alias x (res2 may be
 *   out: in-place running sum).
 *   w1, w2: packed [C][k * C] fp32 (tap-major, as srn_conv_gemm's k-major weights); b1, b2: (C).
 *   w1_hi, w2_hi: SRN_PREC_BF16X3 only -- the same weights as bf16 planes [C][k][C / 32][hi 32 | lo 32].
 */
typedef struct SrnResUnitParams {
  int32_t n_batch, T, C;
  int32_t k, dilation;      /* conv1: kernel k, dilation `dilation`; conv2: kernel k, dilation 1; both "same" padded */
  float slope;              /* LeakyReLU negative slope in front of both convs */
  const float* x; int64_t x_bs;
  const float* w1; const float* b1; const float* w2; const float* b2;
  const void* w1_hi; const void* w2_hi;
  const float* res2; int64_t res2_bs;  /* or NULL */
  float post_div;           /* 0 or 1: none */
  float* out; int64_t out_bs;
  int32_t precision;        /* SRN_PREC_* */
} SrnResUnitParams;

int srn_hifigan_resunit(const SrnResUnitParams* p, void* stream);

/* SiFiGAN pitch-dependent dilated-conv operand gather (row a9; un-vendored `sifigan` package, parity unpinned):
 * out (B, T, 3C) = [lrelu(x[t]) | lrelu(x[t - r]) | lrelu(x[t + r])], r = rint(d[b, t] * dilation), zero outside. */
int srn_pd_gather(const float* x, const float* d, float* out, int B, int T, int C, float dilation, float slope,
                  void* stream);

/* GST style encoder (serenade/modules/gst/style_encoder.py:142-191,235-252).  Its Conv2d(k3, s2, p1) + BatchNorm2d(eval)
 * + ReLU layers are srn_conv_gemm launches (one stride-2 three-tap contraction along the mel axis per kernel row, BatchNorm
 * folded into the weights); the two entry points below are the tail. */
/* torch.nn.GRU's last hidden state (style_encoder.py:169,188-189; gates r, z, n) with the input projection hoisted out:
 * gi (B, T, 3H) = x W_ih^T + b_ih comes from one srn_conv_gemm over all (b, t) rows; this call runs only the recurrence
 * h -> W_hh h.  w_hh_t (H, 3H) = W_hh transposed. */
int srn_gru_recur_last(const float* gi, const float* w_hh_t, const float* b_hh, float* h, int B, int T, int H,
                       void* stream);

/* StyleTokenLayer (style_encoder.py:235-252 + gst/attention.py:110-184,298-300): ref (B, Dq) -> out (B, F), with its
 * input-independent parts formed at weight-packing time: k, v (n_tok, F) = tanh(embs) W_k^T + b_k / W_v^T + b_v;
 * wq_t (Dq, F), wo_t (F, F) = W_q, W_out transposed. */
int srn_style_token_attention_kv(const float* ref, const float* wq_t, const float* bq, const float* k, const float* v,
                                 const float* wo_t, const float* bo, float* out, int B, int Dq, int n_tok, int F,
                                 int n_head, void* stream);

/*
 * Feature front-end in front of the hot path (serenade/bin/preprocess.py:126-203: `loudness_extract`,
 * `logmelfilterbank`; their arithmetic lives in the un-vendored librosa, restated in oracle/features_oracle.py --
 * parity unpinned).  The STFT itself is srn_conv_gemm over the reflect-padded signal viewed as rows of 16 samples with
 * the window x DFT basis as weights; its output rows are [re(0..n_bins-1) | im(0..n_bins-1) | pad] with stride ld.
 */
/* numpy.pad(x, pad, mode) per batch row, zero-filled up to ld: x (B, n) -> out (B, ld).  mode 0 = "reflect" (what
 * logmelfilterbank passes to librosa.stft, preprocess.py:174-181), 1 = "constant" zeros (librosa.stft's default since
 * 0.10, which loudness_extract's call preprocess.py:131 takes). */
int srn_pad_signal(const float* x, float* out, int B, int n, int pad, int ld, int mode, void* stream);
/* out (frames, n_mels) = log_b(max(eps, |spec| @ mel^T)); mel_t (n_bins, n_mels) = filterbank transposed;
 * log_mode 10 / 2 / 0 (natural)  (preprocess.py:176-203). */
int srn_logmel(const float* spec, const float* mel_t, float* out, int64_t frames, int n_bins, int ld, int n_mels,
               float eps, int log_mode, void* stream);
/* out (B, frames) = log(mean_f 10^((max(10 log10(max(amin, |spec|^2)), max_db(utterance) - top_db) + a_weight_db[f])
 * / 20) + add_eps)  (preprocess.py:126-137: power_to_db -> perceptual_weighting -> db_to_amplitude -> mean -> log).
 * gmax_ws: B uint32 of scratch. */
int srn_loudness(const float* spec, const float* a_weight_db, uint32_t* gmax_ws, float* out, int B, int frames,
                 int n_bins, int ld, float amin, float top_db, float add_eps, void* stream);

/*
 * Training step of the estimator: the backward of the decoder blocks that CFM.compute_loss
 * differentiates (flow_matching.py:95-133 -> matcha_components/decoder.py:34-45,66-101,384-467, transformer.py:120-146,
 * 286-352) and the optimizer update (bin/ssc_train.py:331-349, trainers/ssc.py:86-96).  GEMM-shaped gradients (dgrad of
 * convs / projections, dP, dQ) are srn_conv_gemm launches with transposed weights; these are the HBM-bound pieces.
 * Column sums come back as per-row-chunk partial sums the host adds (bit-reproducible, no atomics).
 */
/* y[b,t,:] = LayerNorm_C(x[b,t,:]) * m[b] + a[b]  (no affine inside): nn.LayerNorm with m = gamma, a = beta and batch
 * strides 0; SpeakerAdapter (decoder.py:34-45) with m = W_scale spk + b, a = W_bias spk + b and batch strides C. */
int srn_rowln_fwd(const float* x, const float* m, int64_t m_bs, const float* a, int64_t a_bs, float* y, int B, int T,
                  int C, float eps, void* stream);
/* dx of the above; partial (B, srn_rowln_chunks(T), 2, C): [0] = sum_t dy * xhat (-> dm), [1] = sum_t dy (-> da). */
int srn_rowln_bwd(const float* x, const float* dy, const float* m, int64_t m_bs, float* dx, float* partial, int B, int T,
                  int C, float eps, void* stream);
int srn_rowln_chunks(int T);
/* Block1D's GroupNorm -> Mish -> mask (decoder.py:66-77) backward, statistics over the padded length T.
 * mean, rstd (B, groups); step 1: partial (B, srn_gn_chunks(T), 2, C): [0] = sum_t dg, [1] = sum_t dg * xhat with
 * dg = dy * mish'(gamma xhat + beta) on rows < lens[b]; the host forms d beta, d gamma and gsum (B, groups, 2) =
 * per-group sums of (gamma * [0], gamma * [1]); step 2: dh = rstd (dg gamma - gsum0 / n - xhat gsum1 / n). */
int srn_gn_mish_bwd_partial(const float* h, const float* dy, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, const int32_t* lens, float* partial, int B, int T, int C, int groups,
                            void* stream);
int srn_gn_mish_bwd_apply(const float* h, const float* dy, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, const float* gsum, const int32_t* lens, float* dh, int B, int T, int C,
                          int groups, void* stream);
int srn_gn_chunks(int T);
/* (mean, rstd)[b][g] of GroupNorm from the producing conv's 32 x 32 tile sums (SrnConvParams.gn_partials), statistics
 * over the padded length T: what srn_gn_mish_apply uses internally, kept for the backward pass. */
int srn_gn_stats(const float* partials, float* mean, float* rstd, int B, int T, int C, int groups, float eps,
                 void* stream);
/* col (B, 2, C) = sum over chunks of a (B, n_chunk, 2, C) partial-sum buffer of the kernels above; with gsum != NULL
 * also gsum (B, groups, 2) = per-group sums of gamma * col (step between srn_gn_mish_bwd_partial and _apply). */
int srn_chunk_colsum(const float* partial, const float* gamma, float* col, float* gsum, int B, int n_chunk, int C,
                     int groups, void* stream);
/* out (B, N) = column sums of B row-major (R, N) matrices with row stride ld, batch stride R * ld (bias gradients
 * dY^T 1 with B = 1; the gradient of a per-item broadcast add): two launches, partial (B, srn_colsum_chunks(R), N) is
 * scratch.  Fixed summation order. */
int srn_colsum(const float* x, float* partial, float* out, int B, int64_t R, int N, int ld, void* stream);
int srn_colsum_chunks(int64_t R);
/* softmax backward in place on dp: dp <- scale * p o (dp - rowsum(dp o p)); rows of L with stride ld. */
int srn_softmax_bwd(const float* p, float* dp, int64_t rows, int L, int ld, float scale, void* stream);
/* GEGLU (transformer.py:120-146): hg (rows, 2 inner) = [h | g]; a = h * gelu_erf(g); backward dhg from da. */
int srn_geglu_fwd(const float* hg, float* a, int64_t rows, int inner, void* stream);
int srn_geglu_bwd(const float* hg, const float* da, float* dhg, int64_t rows, int inner, void* stream);
/* torch.optim.AdamW step `step` (1-based) over one flat fp32 buffer: g is scaled by grad_scale first (the
 * clip_grad_norm_ coefficient of trainers/ssc.py:90-94), then p *= 1 - lr wd; m, v updated; p -= lr/bc1 m/(sqrt(v/bc2)+eps). */
int srn_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, float grad_scale, void* stream);
/* The same update with the step-dependent scalars in device memory, dyn = {lr, 1 - beta1^step, 1 - beta2^step,
 * grad_scale}: an identical launch every step, so a training step captured as a hipGraph can replay it. */
int srn_adamw_dyn(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                  float weight_decay, const float* dyn, void* stream);
/* clip_grad_norm_'s total norm: partial[i], i < srn_sumsq_blocks(n) (<= 1024), = fp64 sums of squares of slices of g;
 * the host adds them and takes the root. */
int srn_sumsq(const float* g, int64_t n, double* partial, void* stream);
/* the same with a second operand: partial sums of a . b (b == NULL: of a) -- the loss sums of the training step (a torch
 * reduction of this size zeroes its semaphores with hipMemsetAsync, which a captured hipGraph replays wrongly). */
int srn_dot(const float* a, const float* b, int64_t n, double* partial, void* stream);
/* Many small device-to-device copies in one launch (gradients of 262 parameter tensors into the flat gradient buffer):
 * entry e copies len[e] floats from src[e] to dst + off[e].  The table is passed by value in the kernel arguments. */
#define SRN_COPY_LIST_MAX 160
typedef struct SrnCopyList {
  int32_t n;
  int32_t pad_;
  const void* src[SRN_COPY_LIST_MAX];
  int64_t off[SRN_COPY_LIST_MAX];
  int64_t len[SRN_COPY_LIST_MAX];
} SrnCopyList;
int srn_multi_copy(const SrnCopyList* list, float* dst, void* stream);
int srn_sumsq_blocks(int64_t n);
/* torch.nn.utils.weight_norm of the content encoder's convs (serenade/models/serenade.py `Conv1dResnet`; w = g v / ||v||
 * per output channel) fused with the re-layout the conv kernels want: forward writes the packed k-major rows
 * w[n][j * C + c] = g[n] v[n][c][j] / ||v[n]||, optionally W^T for the input gradient wd[c][j * N + n], and 1 / ||v[n]||;
 * backward takes the packed dW and returns dv (N, C, k) and dg (N,).  One workgroup per output channel, fixed summation
 * order.  (Eleven convs x ~14 torch launches per step otherwise.) */
int srn_weight_norm_fwd(const float* v, const float* g, float* w_packed, float* wd, float* inv_norm, int N, int C, int k,
                        void* stream);
int srn_weight_norm_bwd(const float* dw_packed, const float* v, const float* g, const float* inv_norm, float* dv,
                        float* dg, int N, int C, int k, void* stream);

/*
 * Analysis front-end between HiFi-GAN and SiFiGAN -- SURVEY 8 row f1, serenade/bin/ssc_postprocessing.py:142-222.
 * The reference calls pyworld (`pw.cheaptrick` :167, `pw.d4c` :168, `pw.code_aperiodicity` :171), pysptk
 * (`pysptk.sp2mc` :169) and sifigan.utils.features (`dilated_factor` :201-210, `SignalGenerator` :105-111,219-222):
 * third-party code absent from the reference tree, restated from the published algorithms in oracle/world_oracle.py
 * (parity unpinned).  `convert_continuos_f0` (:51-72) and the np.interp length match (:159-166) are in-tree and pinned.
 * All analysis arithmetic is float64 (WORLD's precision); one workgroup per 5 ms frame, everything in LDS.
 */
typedef struct SrnWorldParams {
  int32_t n_batch, max_frames;  /* grid = max_frames x n_batch; frames >= n_frames[b] are skipped */
  int32_t fs, fft_size;         /* cheaptrick: 512 / 1024 / 2048 (GetFFTSizeForCheapTrick); d4c: 1024 / 2048 */
  const double* x; int64_t x_bs; const int32_t* x_len; /* (B, x_bs) float64 waveforms and their valid lengths */
  const double* f0; const double* t; int64_t f_bs;     /* (B, f_bs) F0 in Hz and temporal positions in seconds */
  const int32_t* n_frames;      /* (B,) */
  const double* twiddle;        /* fft_size/2 pairs (cos, -sin)(2 pi k / fft_size) */
  double q1, f0_floor;          /* cheaptrick: compensation-lifter q1 (-0.15); F0 <= f0_floor is analysed as 500 Hz */
  double threshold;             /* d4c: Love-Train voicing threshold (0.85) */
  double unvoiced_db;           /* d4c: what unvoiced / rejected frames hold, 20 log10(1 - 1e-12) */
  const double* band_window; int32_t band_window_len; /* d4c: Nuttall window over the group delay of one band */
  int32_t n_bands;              /* d4c: 3 kHz bands below min(15 kHz, fs/2 - 3 kHz): 3 at 24 kHz */
  double* out0; int64_t out0_bs; int32_t ld_out0; /* cheaptrick: spectral envelope (fft_size/2+1 per frame) or NULL;
                                                    * d4c: band aperiodicity in dB (n_bands per frame) */
  double* out1; int64_t out1_bs; int32_t ld_out1; /* cheaptrick: liftered cepstrum X (its c2r is log sp) or NULL */
} SrnWorldParams;
/* pw.cheaptrick: F0-adaptive Hanning window of 3 periods, power spectrum, DC correction, smoothing over 2 F0 / 3,
 * cepstral smoothing + q1 compensation lifter. */
int srn_world_cheaptrick(const SrnWorldParams* p, void* stream);
/* pw.d4c followed by pw.code_aperiodicity: Love-Train voicing check, static group delay, band aperiodicity in dB. */
int srn_world_d4c(const SrnWorldParams* p, void* stream);
/* out (rows, n_out) = g(in (rows, K)) @ mat_t (K, n_out), g = log (take_log) or identity, float64: pysptk.sp2mc as one
 * matrix (log -> irfft -> c0 / 2 -> SPTK freqt is linear in log sp), n_out <= 64. */
int srn_world_project(const double* in, int64_t rows, int K, int ld_in, const double* mat_t, int n_out, int take_log,
                      double* out, int ld_out, void* stream);
/* c (rows, ld_out) float32 = ([a | b] - mean) / scale (StandardScaler.transform :185-199, torch.FloatTensor :213);
 * mean = scale = NULL: plain concatenation. */
int srn_world_pack_features(const double* a, int na, const double* b, int nb, const double* mean, const double* scale,
                            float* out, int64_t rows, int ld_out, void* stream);
/* float32 waveform -> float64 samples; pcm16: through the PCM_16 file hop of the reference (decode writes
 * lrint(x * 32767), ssc_decode.py:449-455; post-processing reads int16 / 32768, ssc_postprocessing.py:143). */
int srn_wave_to_f64(const float* wave, double* out, int64_t n, int pcm16, void* stream);
/* np.maximum(np.interp(np.linspace(0, n_in - 1, n_out), arange(n_in), f0), 0); copy when n_in == n_out (:159-166). */
int srn_f0_match_length(const double* in, int64_t in_bs, const int32_t* n_in, double* out, int64_t out_bs,
                        const int32_t* n_out, int n_batch, int max_out, void* stream);
/* convert_continuos_f0 (:51-72): cf0 float64, uv float32 (both (B, bs)), ok[b] = 0 when an item has no voiced frame. */
int srn_cont_f0(const double* f0, int64_t bs, const int32_t* n_frames, double* cf0, float* uv, int32_t* ok,
                int n_batch, void* stream);
/* SignalGenerator(signal_types=["sine"]) and the dilated-factor tracks np.repeat(dilated_factor(f0, fs, dense), us). */
typedef struct SrnExcitationParams {
  int32_t n_batch, max_frames, fs, hop;
  const double* f0;        /* (B, f_bs) contour driving the sine (cf0 or f0, `sine_f0_type`) */
  const double* df_f0;     /* (B, f_bs) contour driving the dilated factors (`df_f0_type`) */
  int64_t f_bs;
  const int32_t* n_frames;
  double* phase_ws;        /* (B, f_bs) scratch: phase at every frame start */
  const float* noise;      /* (B, max_frames * hop) standard-normal draws standing in for torch.randn, or NULL */
  float* sine;             /* (B, max_frames * hop) */
  float sine_amp, noise_amp;
  int32_t n_df;            /* <= 4 */
  float* dfs[4];           /* track i: (B, max_frames * df_upsample[i]) */
  int32_t df_upsample[4];  /* np.cumprod(upsample_scales) */
  double dense_factors[4];
} SrnExcitationParams;
int srn_sifigan_excitation(const SrnExcitationParams* p, void* stream);

/*
 * Contractions over TIME of the training step (what `loss.backward()` of trainers/ssc.py:57-96 runs for the
 * Conv1d / Linear weights of matcha_components/decoder.py and for diffusers' attention), exact fp32 MFMA:
 *
 *   out[z, m, j * N + n] = alpha * sum_{item, t} a[z, item, t, m] * b[z, item, t * stride + shift[j], n]
 *
 * with rows of b outside [0, T_b) -- or [0, len_b[item]) -- read as zero.  z = zb * n_head + zh walks independent problems.
 *   weight gradient of a conv:  a = dY (B items of T_out rows, M = C_out), b = X (T_in rows, N = C_in), shift = taps
 *                               -> dW in the packed layout (C_out, taps * C_in) srn_conv_gemm reads;
 *   attention:  dV = P^T dO and dK = dS^T Q, one problem per (batch, head), n_items = 1, T_a = T_b = L.
 * Operands are used as they lie in memory (time-major): no transposes, no padded copies.  When the output has too
 * few tiles to fill the chip the time axis is sliced over extra workgroups; the slices go through `ws`
 * (srn_tn_gemm_workspace_bytes) and are added in slice order: results are bit-reproducible.
 */
typedef struct SrnTnGemmParams {
  int32_t n_batch, n_head;   /* problems */
  int32_t n_items, T_a, T_b; /* contraction: items x T_a rows of a, paired with rows of b in [0, T_b) */
  int32_t stride, n_shifts;
  int32_t shift[SRN_MAX_TAPS];
  int32_t M, N;              /* N % 4 == 0; M free with lda >= roundup(M, 4) */
  const float* a; int64_t a_bs, a_hs, a_is; int32_t lda;
  const float* b; int64_t b_bs, b_hs, b_is; int32_t ldb;
  float* out; int64_t out_bs, out_hs; int32_t ldc;  /* ldc >= n_shifts * N */
  float alpha;
  float* ws; int64_t ws_bytes;  /* or NULL: no slicing */
  int32_t n_inner;              /* > 1: item i = (i / n_inner, i % n_inner) with strides (x_is, x_is2) -- conv2d's */
  int64_t a_is2, b_is2;         /* (batch, output row) items; 0 / 1: one level */
  const int32_t* len_b;         /* or NULL; (n_batch, n_items), one-level items only: rows of b at or past len_b[zb, item]
                                 * read as zero -- the `x * mask` in front of the reference's convs (decoder.py:66-101),
                                 * as srn_conv_gemm's len_in does it in the forward */
  float* colsum;                /* or NULL; (M,), one problem, M % 4 == 0: alpha * sum_{item, t} a[item, t, m] -- a
                                 * conv's bias gradient (the column sums of dY), added in the same slice order */
} SrnTnGemmParams;
int srn_tn_gemm(const SrnTnGemmParams* p, void* stream);
int64_t srn_tn_gemm_workspace_bytes(const SrnTnGemmParams* p);

/*
 * Training-mode pieces of the GST style encoder (serenade/modules/gst/style_encoder.py:171-191,235-252 under autograd;
 * SURVEY 8 f4).  Channels-last rows throughout.
 */
/* BatchNorm2d(training) + ReLU over `rows` x C: stats (2, C) = (batch mean, 1 / sqrt(biased var + eps)) kept for the
 * backward; running statistics updated like nn.BatchNorm2d (momentum, unbiased variance) when given.
 * partial: srn_bn_chunks(rows) * 2 * C floats of scratch.  C % 4 == 0. */
int srn_bn_chunks(int64_t rows);
int srn_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var,
                    float* partial, float* stats, float* y, int64_t rows, int C, float eps, float momentum, void* stream);
/* dx, and sums (2, C) = (dbeta, dgamma). */
int srn_bn_relu_bwd(const float* x, const float* y, const float* dy, const float* stats, const float* gamma,
                    float* partial, float* sums, float* dx, int64_t rows, int C, void* stream);
/* Conv2d(k 3, stride 2, pad 1) on channels-last (B, H, W, C) as a plain GEMM: col[(b, ho, wo), (kh, kw, c)] =
 * x[b, 2 ho + kh - 1, 2 wo + kw - 1, c] (zero outside), row stride ld >= 9 C (pad columns are not written).  The
 * per-(b, ho) form of the inference path leaves 64-row tiles mostly empty once Wo <= 20; flattened, every layer is
 * one well-shaped contraction per direction.  srn_col2im_s2 is the transpose of the gather (the dX side). */
int srn_im2col_s2(const float* x, float* col, int B, int H, int W, int C, int ld, void* stream);
int srn_col2im_s2(const float* col, float* dx, int B, int H, int W, int C, int ld, void* stream);
/* nn.GRU recurrence (gate order r, z, n) on gi = x W_ih^T + b_ih (B, T, 3H), zero initial state: hs (B, T+1, H) all
 * hidden states, gates (B, T, 4H) = [r | z | n | W_hn h + b_hn].  w_hh_t = W_hh transposed (H, 3H). */
int srn_gru_train_fwd(const float* gi, const float* w_hh_t, const float* b_hh, float* hs, float* gates, int B, int T,
                      int H, void* stream);
/* BPTT from dh_last (B, H): dgi, dgh (B, T, 3H); dW_hh = dgh^T hs[:, :T], db_hh = sum dgh are srn_tn_gemm / srn_colsum. */
int srn_gru_train_bwd(const float* dh_last, const float* w_hh, const float* hs, const float* gates, float* dgi,
                      float* dgh, int B, int T, int H, void* stream);
/* StyleTokenLayer's attention core: one query row q (B, F) per item over keys / values (n_tok, F), n_head heads:
 * p (B, n_head, n_tok) softmax weights, ctx (B, F). */
int srn_token_attn_fwd(const float* q, const float* k, const float* v, float* p, float* ctx, int B, int n_tok, int F,
                       int n_head, void* stream);
/* dq (B, F); dk_part, dv_part (B, n_tok, F): per-item terms, summed over items by the caller (srn_colsum). */
int srn_token_attn_bwd(const float* dctx, const float* q, const float* k, const float* v, const float* p, float* dq,
                       float* dk_part, float* dv_part, int B, int n_tok, int F, int n_head, void* stream);

#ifdef __cplusplus
}
#endif
#endif
