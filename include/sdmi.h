/* libsdmi -- C ABI of the MI355X-native Stable-Diffusion-v1 denoising hot path.
 *
 * The reference (ShaShekhar/Stable-diffusion) is pure Python: the path it runs per DDIM step is
 * nn.Module.forward calls, not an FFI.  This header is therefore the boundary a maintainer binds
 * with ctypes underneath those modules; every entry point names the reference code it replaces
 * (file = stable_diffusion.py unless stated).
 *
 * Conventions
 *   - plain C: pointers and sizes only; no torch types, no exceptions across the boundary;
 *   - every function returns 0 on success or a negative code (SDMI_E*); the message for the calling
 *     thread's last failure is returned by sdmi_last_error();
 *   - all pointers named x/out/ctx/t/... are DEVICE pointers owned by the caller; the library
 *     never frees them.  Weights are copied / repacked into handle-owned device memory;
 *   - work is enqueued asynchronously on the HIP stream passed as `stream` (a hipStream_t cast to
 *     void*; NULL = the default stream).  One handle must be used from one host thread at a time;
 *     one handle per device.
 *   - dtype codes: 0 = fp32, 1 = fp16, 2 = bf16.
 */
#ifndef SDMI_H
#define SDMI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDMI_VERSION 100          /* 0.1.0 */

#define SDMI_DTYPE_F32 0
#define SDMI_DTYPE_F16 1
#define SDMI_DTYPE_BF16 2

#define SDMI_OK 0
#define SDMI_EINVAL (-22)
#define SDMI_ENOMEM (-12)
#define SDMI_EHIP (-5)
#define SDMI_ENOENT (-2)
#define SDMI_ESTATE (-1)

int sdmi_version(void);
const char* sdmi_last_error(void);

/* ------------------------------------------------------------------------------------------
 * UNet denoiser: replaces UNetModel.__init__ (:3263-3496) / UNetModel.forward (:3519-3552) and
 * the DiffusionWrapper routing in front of it (:3646-3664).
 * ------------------------------------------------------------------------------------------ */
typedef struct sdmi_unet sdmi_unet;

typedef struct sdmi_unet_cfg {
  int in_channels;               /* 4, or 9 for the hybrid inpainting model            */
  int model_channels;            /* 320                                                */
  int out_channels;              /* 4                                                  */
  int num_res_blocks;            /* 2                                                  */
  int n_channel_mult;
  int channel_mult[8];           /* 1,2,4,4                                            */
  int n_attention_resolutions;
  int attention_resolutions[8];  /* 4,2,1                                              */
  int num_heads;                 /* 8, or -1 when num_head_channels is used            */
  int num_head_channels;         /* -1, or channels per head                           */
  int transformer_depth;         /* 1                                                  */
  int context_dim;               /* 768                                                */
  int legacy;                    /* the reference's `legacy` flag (:3352)              */
  int compute_dtype;             /* activations / MFMA operand type; fp32 = parity mode */
} sdmi_unet_cfg;

int sdmi_unet_create(const sdmi_unet_cfg* cfg, sdmi_unet** out);
/* Number of checkpoint tensors the configuration expects, and the i-th key / shape (relative to
 * "model.diffusion_model.").  Replaces nn.Module.state_dict() introspection. */
int sdmi_unet_num_weights(const sdmi_unet* u);
int sdmi_unet_weight_info(const sdmi_unet* u, int index, const char** key, int64_t* shape4, int* ndim);
/* Copy one checkpoint tensor (device pointer, fp32, reference layout: OIHW convs, (out,in)
 * linears) into the handle.  Replaces load_state_dict (txt2img.py:213-220). */
int sdmi_unet_set_weight(sdmi_unet* u, const char* key, const void* dev_ptr, const int64_t* shape,
                         int ndim, int dtype, void* stream);
/* Repack all weights to kernel layouts; fails with SDMI_ENOENT (message lists keys) if any is missing. */
int sdmi_unet_finalize(sdmi_unet* u, void* stream);
/* Project the text context once per prompt batch: K/V of every cross-attention block
 * (CrossAttention.to_k/to_v, :2781-2782) are cached in the handle and reused by every step.
 * ctx: fp32 [batch][ntok][context_dim]. */
int sdmi_unet_set_context(sdmi_unet* u, const void* ctx, int batch, int ntok, void* stream);
/* eps = UNet(cat([x] + concat, 1), t, context)   (:3519; hybrid cat :3656)
 * x: fp32 NCHW [batch][c_x][h][w]; concat[i]: fp32 NCHW [batch][concat_channels[i]][h][w] (n_concat
 * <= 2, c_x + sum = in_channels); t: int64 [batch]; out: fp32 NCHW [batch][out_channels][h][w].
 * The context set by sdmi_unet_set_context must have the same batch. */
int sdmi_unet_forward(sdmi_unet* u, const void* x, const void* const* concat, const int* concat_channels,
                      int n_concat, const int64_t* t, void* out, int batch, int h, int w, void* stream);
/* One block of the denoiser on its own, through the code path sdmi_unet_forward uses for it: replaces
 * ResBlock.forward(x, emb) :3197-3234 (kind 0) and SpatialTransformer.forward(x, context) :2885-2896 (kind 1,
 * context = the one given to sdmi_unet_set_context).  index counts the blocks of that kind in construction
 * order (ResBlocks: input_blocks.1.0, input_blocks.2.0, ...; transformers: input_blocks.1.1, ...).
 * x, out: NHWC [batch][h][w][channels] in the handle's compute dtype; x_skip: optional second input that is
 * channel-concatenated behind x (the skip tensor of an output block), c_skip its channels; emb: fp32
 * [batch][4 * model_channels], the output of time_embed (the block applies its own SiLU + Linear). */
int sdmi_unet_run_block(sdmi_unet* u, int kind, int index, const void* x, const void* x_skip, int c_skip,
                        const float* emb, void* out, int batch, int h, int w, void* stream);
/* 1: the caller promises that every sample of the following forwards carries the same timestep (the
 * DDIM loop's torch.full((b,), step), :5308): the time-embedding MLP then runs on t[0] only.  Default 0. */
int sdmi_unet_set_uniform_timesteps(sdmi_unet* u, int enable);
/* 1: the caller promises that in the following forwards sample b and sample b + batch/2 have the same x
 * and timestep and differ only in their context rows (p_sample_ddim's x_in = cat([x] * 2), :5347-5349):
 * the layers ahead of the first cross-attention then run on one half of the batch.  Ignored for hybrid
 * (concat) inputs and odd batches.  Default 0. */
int sdmi_unet_set_cfg_duplicate(sdmi_unet* u, int enable);
/* Bytes of device memory currently held (weights + caches + workspace). */
int64_t sdmi_unet_device_bytes(const sdmi_unet* u);
/* 1: replay the forward as a captured HIP graph when shapes and pointers repeat (default), 0: eager. */
int sdmi_unet_set_graph_mode(sdmi_unet* u, int enable);
/* Kernel launches in the currently recorded forward graph (0 while none is recorded): what one replayed
 * UNetModel.forward :3519-3552 costs in launches. */
int sdmi_unet_graph_launches(const sdmi_unet* u);
void sdmi_unet_destroy(sdmi_unet* u);

/* ------------------------------------------------------------------------------------------
 * Sampler arithmetic: replaces DDIMSampler.p_sample_ddim :5344-5347, :5360-5375 (CFG combine +
 * DDIM update), stochastic_encode :5378-5391 and DDPM.q_sample :3892-3895.  All fp32.
 * ------------------------------------------------------------------------------------------ */
/* eps: [e_uncond | e_cond] (2n floats) when cfg != 0, else n floats.  noise may be NULL (sigma 0).
 * pred_x0 may be NULL. */
int sdmi_ddim_step(const void* x, const void* eps, int cfg, float scale, float a_t, float a_prev,
                   float sigma, float sqrt_one_minus_at, const void* noise, float temperature,
                   void* x_prev, void* pred_x0, int64_t n, void* stream);
/* DDPM ancestral update, replaces LatentDiffusion.p_mean_variance :4672-4684 (predict_start_from_noise
 * :3821-3825, the optional clamp, q_posterior :3839-3846) and the last line of p_sample :4710-4721:
 *   x0     = sqrt_recip_acp * x - sqrt_recipm1_acp * eps          (clamped to [-1,1] if clip_denoised)
 *   x_prev = mean_coef1 * x0 + mean_coef2 * x + noise_coef * (noise * temperature)
 * The scalars are the schedule tables' entries at the step's timestep; noise_coef is
 * (t != 0) * exp(0.5 * posterior_log_variance_clipped[t]).  noise may be NULL (t == 0); pred_x0 may be NULL. */
int sdmi_ddpm_step(const void* x, const void* eps, float sqrt_recip_acp, float sqrt_recipm1_acp,
                   int clip_denoised, float mean_coef1, float mean_coef2, float noise_coef, const void* noise,
                   float temperature, void* x_prev, void* pred_x0, int64_t n, void* stream);
/* Known-region blend of the DDIM loop, replaces DDIMSampler.ddim_sampling :5316-5319 with DDPM.q_sample
 * :3892-3895 inlined:  out = (sqrt_acp * x0 + sqrt_one_minus_acp * noise) * mask + (1 - mask) * img.
 * The two scalars are sqrt_alphas_cumprod[t] / sqrt_one_minus_alphas_cumprod[t] of the step's timestep (one
 * t for the whole batch, :5314).  x0, noise, img, out: fp32 [batch][channels][hw]; mask: fp32
 * [batch][mask_channels][hw], mask_channels 1 (broadcast over channels) or `channels`.  out may alias img. */
int sdmi_q_sample_blend(const void* x0, const void* noise, const void* mask, const void* img, float sqrt_acp,
                        float sqrt_one_minus_acp, void* out, int batch, int channels, int hw, int mask_channels,
                        void* stream);
/* The guidance combine of p_sample_ddim :5347 alone: out = e_uncond + scale * (e_cond - e_uncond), eps =
 * [e_uncond | e_cond] (2n floats).  For callers that modify the combined score (score_corrector, :5351-5353)
 * before the update; sdmi_ddim_step with cfg = 0 then takes the result. */
int sdmi_cfg_combine(const void* eps, float scale, void* out, int64_t n, void* stream);
/* out = ca * a + cb * b */
int sdmi_axpby(const void* a, const void* b, float ca, float cb, void* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * VAE decoder: replaces LatentDiffusion.decode_first_stage :4293-4346 -> AutoencoderKL.decode
 * :1416-1419 -> Decoder.forward :496-527.
 * ------------------------------------------------------------------------------------------ */
typedef struct sdmi_vae_decoder sdmi_vae_decoder;

typedef struct sdmi_vae_cfg {
  int z_channels;        /* 4   */
  int embed_dim;         /* 4   */
  int ch;                /* 128 (ddconfig 'ch_out') */
  int out_ch;            /* 3   */
  int n_ch_mult;
  int ch_mult[8];        /* 1,2,4,4 */
  int num_res_blocks;    /* 2   */
  int compute_dtype;
} sdmi_vae_cfg;

int sdmi_vae_decoder_create(const sdmi_vae_cfg* cfg, sdmi_vae_decoder** out);
int sdmi_vae_decoder_num_weights(const sdmi_vae_decoder* v);
int sdmi_vae_decoder_weight_info(const sdmi_vae_decoder* v, int index, const char** key, int64_t* shape4, int* ndim);
int sdmi_vae_decoder_set_weight(sdmi_vae_decoder* v, const char* key, const void* dev_ptr,
                                const int64_t* shape, int ndim, int dtype, void* stream);
int sdmi_vae_decoder_finalize(sdmi_vae_decoder* v, void* stream);
/* out = Decoder(post_quant_conv(z * z_scale)); z fp32 NCHW [batch][z_channels][h][w],
 * out fp32 NCHW [batch][out_ch][8h][8w].  z_scale = 1/scale_factor for decode_first_stage, 1 for
 * AutoencoderKL.decode.  Any h, w (the mid-block attention runs on the matrix cores when h * w is a multiple of 64,
 * 32 in fp32 mode, which every latent behind the UNet is, and on the vector ALU otherwise). */
int sdmi_vae_decoder_forward(sdmi_vae_decoder* v, const void* z, float z_scale, void* out, int batch,
                             int h, int w, void* stream);
void sdmi_vae_decoder_destroy(sdmi_vae_decoder* v);
/* VAE encoder (AutoencoderKL.encode :1410-1414, Encoder.forward :393-420, DownsampleEncoder :88-99).
 * Same cfg struct: out_ch is the number of image channels; the weights are `encoder.*` and `quant_conv.*`.
 * x: fp32 NCHW [batch][out_ch][h][w] (h, w divisible by 2^(n_ch_mult-1)); moments: fp32 NCHW
 * [batch][2 embed_dim][h/8][w/8] = (mean | logvar) of the posterior, the argument of
 * DiagonalGaussianDistribution :624. */
typedef struct sdmi_vae_encoder sdmi_vae_encoder;
int sdmi_vae_encoder_create(const sdmi_vae_cfg* cfg, sdmi_vae_encoder** out);
int sdmi_vae_encoder_num_weights(const sdmi_vae_encoder* v);
int sdmi_vae_encoder_weight_info(const sdmi_vae_encoder* v, int index, const char** key, int64_t* shape4, int* ndim);
int sdmi_vae_encoder_set_weight(sdmi_vae_encoder* v, const char* key, const void* dev_ptr,
                                const int64_t* shape, int ndim, int dtype, void* stream);
int sdmi_vae_encoder_finalize(sdmi_vae_encoder* v, void* stream);
int sdmi_vae_encoder_forward(sdmi_vae_encoder* v, const void* x, void* moments, int batch, int h, int w,
                             void* stream);
void sdmi_vae_encoder_destroy(sdmi_vae_encoder* v);
/* z = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise)   (DiagonalGaussianDistribution.sample :636-638,
 * get_first_stage_encoding :4264-4271 with scale = scale_factor); noise NULL: z = scale * mean (mode :656).
 * moments fp32 [batch][2 channels][hw], noise / z fp32 [batch][channels][hw]. */
int sdmi_gaussian_sample(const void* moments, const void* noise, float scale, void* z, int batch, int channels,
                         int hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * CLIP text transformer: replaces the arithmetic of FrozenCLIPEmbedder.forward :2569-2576
 * (transformers' CLIPTextModel(input_ids).last_hidden_state; the tokenizer stays on the host).
 * Weights use the checkpoint key names below `cond_stage_model.transformer.`:
 * text_model.embeddings.{token,position}_embedding.weight, text_model.encoder.layers.N.
 * {self_attn.{q,k,v,out}_proj, layer_norm1, mlp.fc1, mlp.fc2, layer_norm2}.{weight,bias},
 * text_model.final_layer_norm.{weight,bias}.
 * ------------------------------------------------------------------------------------------ */
typedef struct sdmi_clip_text sdmi_clip_text;
typedef struct sdmi_clip_cfg {
  int vocab_size;     /* 49408 */
  int hidden;         /* 768   */
  int intermediate;   /* 3072  */
  int layers;         /* 12    */
  int heads;          /* 12    */
  int max_positions;  /* 77    */
  float ln_eps;       /* 1e-5  */
  int compute_dtype;
} sdmi_clip_cfg;
int sdmi_clip_text_create(const sdmi_clip_cfg* cfg, sdmi_clip_text** out);
int sdmi_clip_text_num_weights(const sdmi_clip_text* c);
int sdmi_clip_text_weight_info(const sdmi_clip_text* c, int index, const char** key, int64_t* shape4, int* ndim);
int sdmi_clip_text_set_weight(sdmi_clip_text* c, const char* key, const void* dev_ptr, const int64_t* shape,
                              int ndim, int dtype, void* stream);
int sdmi_clip_text_finalize(sdmi_clip_text* c, void* stream);
/* token_ids: device int64 [batch][tokens] (tokens <= max_positions, causal attention, no padding mask, as
 * the reference calls the model); out: fp32 [batch][tokens][hidden] = last_hidden_state. */
int sdmi_clip_text_forward(sdmi_clip_text* c, const int64_t* token_ids, void* out, int batch, int tokens,
                           void* stream);
void sdmi_clip_text_destroy(sdmi_clip_text* c);

/* out = clamp((x + 1) / 2, 0, 1)   (txt2img.py:307) */
int sdmi_to_image(const void* x, void* out, int64_t n, void* stream);
/* out = uint8(255 * clamp((x + 1) / 2, 0, 1)), x fp32 NCHW [batch][channels][hw] -> out uint8 NHWC
 * (txt2img.py:307, :316-317: the array handed to Image.fromarray) */
int sdmi_to_image_u8(const void* x, void* out, int batch, int channels, int hw, void* stream);
/* out = (1 - mask) * clamp((image + 1) / 2, 0, 1) + mask * predicted   (inpaint_with_text.py:286-291);
 * image, predicted, out fp32 NCHW [batch][channels][hw], mask fp32 [batch][1][hw] */
int sdmi_inpaint_composite(const void* image, const void* mask, const void* predicted, void* out, int batch,
                           int channels, int hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-op entry points (unit parity tests; also usable on their own).  Activations are NHWC /
 * token-major in the given dtype unless stated.
 * ------------------------------------------------------------------------------------------ */
/* GroupNorm32 :2266 / Normalize :32 (+ optional SiLU); x1 may be NULL; scratch: fp32 device buffer
 * of sdmi_groupnorm_scratch_bytes(batch, groups) bytes (the two launches of a large tensor exchange their
 * partial sums through it; a tensor whose (sample, group) slices fit a workgroup's registers is normalised
 * in one launch that leaves it untouched). */
int64_t sdmi_groupnorm_scratch_bytes(int batch, int groups);
int sdmi_groupnorm(const void* x0, int c0, const void* x1, int c1, int batch, int hw, int groups, float eps,
                   const float* gamma, const float* beta, int silu, void* out, void* scratch, int dtype,
                   void* stream);
/* sdmi_groupnorm for callers that cannot allocate or bring a workspace: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments and the
 * same results bit for bit, minus `scratch`: the partial sums the two launches of a large tensor exchange live in a buffer
 * built into the library (a device array of 2 MiB + 8 KiB, one copy per GPU, held until the process ends),
 * which limits a call to batch <= 32 and groups <= 32; larger calls fail with SDMI_EINVAL and a message.
 * The call makes no hipMalloc / hipFree and calls no allocating routine.  All calls on one GPU share that
 * GPU's buffer: calls that overlap in time on one GPU (different streams) race; calls on one stream are safe.
 * Nothing in the buffer needs clearing between calls (the first launch writes every word the second reads).
 * The buffer's address on a GPU is looked up once per GPU: by the finalize call of any model handle on that
 * GPU, otherwise by the first call of this entry point there, which must then happen outside stream capture
 * (later calls may be captured). */
int sdmi_groupnorm_static(const void* x0, int c0, const void* x1, int c1, int batch, int hw, int groups, float eps,
                          const float* gamma, const float* beta, int silu, void* out, int dtype, void* stream);
/* nn.LayerNorm (:2807-2809) over the last dim */
int sdmi_layernorm(const void* x, void* out, const float* gamma, const float* beta, int rows, int d, float eps,
                   int dtype, void* stream);
/* Linear(LayerNorm(x)) as ONE pass over x: the GEMM reads the raw rows, the weights carry gamma and the
 * epilogue applies mean, rstd and W beta (y = rstd (x (W diag gamma)^T - mean colsum) + b + W beta).  Replaces
 * the pairs norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q and norm3 -> ff.net[0].proj of BasicTransformerBlock
 * (:2815-2819, :2780-2782, :2703).  x: [rows][k] in dtype; w: fp32 (n, k) in checkpoint layout; bias may be NULL;
 * geglu = 1: w is the (2F, k) GEGLU projection and out is [rows][F] = value * gelu(gate); splits as in
 * sdmi_conv2d.  workspace: sdmi_layernorm_linear_workspace_bytes(...) bytes of device memory (the UNet handle
 * folds and packs once at load and keeps only the per-row statistics in its workspace). */
int64_t sdmi_layernorm_linear_workspace_bytes(int rows, int k, int n, int splits, int dtype);
int sdmi_layernorm_linear(const void* x, const float* w, const float* bias, const float* gamma, const float* beta,
                          float eps, void* out, int rows, int k, int n, int geglu, int splits, void* workspace,
                          int dtype, void* stream);
/* Pack an OIHW (or (out,in) with taps = 1) fp32 weight to the conv/linear kernel layout. */
int sdmi_pack_weight(const float* w, void* out, int out_ch, int in_ch, int taps, int dtype, void* stream);
int sdmi_pack_geglu(const float* w, const float* b, void* wout, float* bout, int inner, int in_ch, int dtype,
                    void* stream);
/* nn.Conv2d 3x3 pad 1 / 1x1 / nn.Linear as implicit GEMM (Downsample :3095 stride 2, Upsample
 * :3062-3064 up = 1, skip-concat :3545 via a1; pad_lo = 1, or 0 for DownsampleEncoder :96-99's
 * F.pad (0,1,0,1) + stride-2 conv).  w from sdmi_pack_weight.  splits >= 1; partial =
 * fp32 scratch of splits*M*N floats when splits > 1. */
int sdmi_conv2d(const void* a0, int c0, const void* a1, int c1, const void* w, const float* bias,
                const float* rowadd, const void* resid, int ldr, void* out, int ldo, int batch, int hin,
                int win, int taps, int stride, int pad_lo, int up, int n_out, int geglu, int splits,
                void* partial, int dtype, void* stream);
/* The convolutions on the NCHW fp32 boundary.  conv_in: the first 3x3 convolution of the UNet (input_blocks.0.0,
 * :3328-3331; with the hybrid concat of DiffusionWrapper :3656 folded in) and of the VAE (conv_in :363, :468):
 * src[i] fp32 NCHW [batch][channels[i]][h][w], channel-concatenated in order (n_src <= 3, at most 64 channels in
 * all); w fp32 OIHW [cout][cin][3][3]; out NHWC [batch][h][w][cout] in dtype.  conv_out: the last 3x3 convolution
 * (out.2 :3491-3495, conv_out :413, :522): x NHWC in dtype, out fp32 NCHW [batch][cout][h][w], cout in
 * {3, 4, 8} (any cout <= 16 when cin is 128 or 320).  wpack: fp32 scratch of 9 * cin * cout floats (the
 * handles pack once at load). */
int sdmi_conv_in(const void* const* src, const int* channels, int n_src, const float* w, const float* bias,
                 float* wpack, void* out, int batch, int h, int width, int cout, int dtype, void* stream);
int sdmi_conv_out(const void* x, const float* w, const float* bias, float* wpack, float* out, int batch, int h,
                  int width, int cin, int cout, int dtype, void* stream);
/* Split count the UNet executor uses for an (m x k) * (k x n) problem of this dtype (1 = no split);
 * callers that want the executor's schedule size `partial` from it.  No counterpart in the reference. */
int sdmi_conv2d_splits(int m, int n, int k, int dtype);
/* softmax(q k^T * scale) v per (batch, head)  (CrossAttention :2784-2796).  impl is a bit set:
 * 1 = force the scalar fp32-math kernel, 2 = causal (query i sees keys 0..i: the text transformer's mask).
 * Every dtype runs on the matrix cores for head sizes 40, 64, 80, 128, 160 (fp32: fp32 MFMA, 16-byte aligned
 * pointers and row strides that are multiples of 4 elements; anything else takes the scalar kernel).
 * Row strides are multiples of 8 elements for the 16-bit kernels, and one batch element's K (and V) must span
 * less than 2 GiB (nk * ldk * 2 bytes): the matrix-core kernel addresses them through 32-bit buffer offsets.
 * (Head size 40 with at most 128 keys and 2048 or more 128-query blocks in all runs a variant that stages K/V once
 * per workgroup; its results equal the general kernel's bit for bit.) */
int sdmi_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo,
                   int batch, int heads, int nq, int nk, int d, float scale, int dtype, int impl, void* stream);
/* timestep_embedding :2212-2232 (fp32 out [batch][dim]) */
int sdmi_timestep_embedding(const int64_t* t, float* out, int batch, int dim, float max_period, void* stream);
/* y = act_out(act_in(x) W^T + b), fp32, small M (time_embed :3317-3322, emb_layers :3180-3183) */
int sdmi_small_linear(const float* x, const float* w, const float* bias, float* y, int m, int k, int n,
                      int act_in, int act_out, void* stream);
/* dtype conversion helpers for tests */
int sdmi_convert_from_f32(const float* x, void* out, int64_t n, int dtype, void* stream);
int sdmi_convert_to_f32(const void* x, float* out, int64_t n, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (no counterpart in the reference, which has no profiling): when enabled, every
 * eager launch of a kernel class is bracketed by HIP events on its stream and the algorithmic
 * FLOPs / bytes of the launch are tallied.  Classes: "conv_gemm", "attention", "groupnorm",
 * "layernorm".  Disable HIP-graph replay (sdmi_unet_set_graph_mode(u, 0)) while profiling.
 * ------------------------------------------------------------------------------------------ */
int sdmi_prof_enable(int on);
int sdmi_prof_reset(void);
int sdmi_prof_get(const char* kernel_class, double* ms_total, int64_t* launches, double* flops, double* bytes);
/* Write one CSV line per profiled launch (class, shape label, microseconds, flops, bytes). */
int sdmi_prof_dump(const char* path);
/* Mean elapsed time (microseconds) of n empty event brackets on `stream`: the part of a profiled
 * launch's bracket that is not the kernel. */
int sdmi_prof_calibrate(int n, void* stream, double* us_per_bracket);

#ifdef __cplusplus
}
#endif
#endif /* SDMI_H */
