/* libcfdenoise -- C ABI of the MI355X-native ConvoFusion denoising loop.
 *
 * The reference (m-hamza-mughal/convofusion) is pure Python/PyTorch and has no FFI; its plug point
 * for this path is `instantiate_from_config` on a dotted class path (convofusion/config.py:16-31,
 * configs/modules/denoiser.yaml:2, configs/modules/scheduler.yaml:2).  This header is the boundary a
 * replacement binds underneath that plug point: plain pointers and sizes, no torch types.  Each
 * entry point names the reference interface it replaces.  All `dev` pointers are device (HBM)
 * pointers, e.g. `tensor.data_ptr()`; the caller owns every buffer it passes.  A handle may be used
 * from one host thread at a time.  Every function returns 0 on success or a negative CFD_E_* code;
 * cfd_last_error() then describes the failure (the reference raises Python exceptions instead:
 * TypeError / ValueError / torch shape errors, denoiser.py:113,123,171,280).
 */
#ifndef CFDENOISE_H
#define CFDENOISE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFD_OK 0
#define CFD_E_ARG (-1)      /* bad argument / unsupported configuration (reference: TypeError/ValueError) */
#define CFD_E_SHAPE (-2)    /* shape the reference also rejects: odd L, L/2 or S beyond the PE buffers */
#define CFD_E_STATE (-3)    /* call order (weights not finalized, no sampling run open, ...) */
#define CFD_E_HIP (-4)      /* HIP runtime error */
#define CFD_E_RANGE (-5)    /* a weight, a centred memory row or a folded key / value projection of one exceeds +-65504, the range of the fp16
                               split-pair operands: the engine would clamp it silently (the reference is float32 and has no such limit);
                               rescale the offending conditioning input */

#define CFD_NUM_MEM 5       /* memory tuple order: spkemb, alsn, tlsn, apb, lsnemb (denoiser.py:220) */

typedef struct cfd_handle_s* cfd_handle;

/* Replaces Denoiser.__init__ (convofusion/models/architectures/denoiser.py:18-171) for the shipped
 * configuration (configs/modules/denoiser.yaml): condition text+audio, arch trans_dec, pre-norm, gelu,
 * sine PEs.  Dimensions other than the ones below are rejected with CFD_E_ARG. */
typedef struct {
  int latent_dim;        /* 128  (latent_dim[-1]) */
  int text_encoded_dim;  /* 512 */
  int ff_size;           /* 1024 */
  int num_layers;        /* 9 (1..16 accepted) */
  int num_heads;         /* 4 */
  int device;            /* HIP device ordinal */
} cfd_config;

int cfd_create(const cfd_config* cfg, cfd_handle* out);
void cfd_destroy(cfd_handle h);
const char* cfd_last_error(void);
/* Hash of the sources this library was built from (convofusion_amd/build.py: source_hash).  The ctypes binding compares
 * it with the sources on disk before the first call: a library left over from other sources -- e.g. after an update that
 * changed a struct of this header -- is rebuilt or refused instead of mis-reading its arguments. */
const char* cfd_source_hash(void);

/* Replaces nn.Module.load_state_dict for the `denoiser.*` entries of the checkpoint
 * (layout: SURVEY.md section 8b; test.py:109-111 -> base.py:106-123).  `name` is the key without the
 * `denoiser.` prefix, e.g. "decoder.layers.0.self_attn.in_proj_weight"; `data` is float32, `numel`
 * elements, on the host (is_device = 0) or the device (1).  Buffers `query_pos.pe` / `mem_pos.pe`
 * may be longer than the checkpoint's 1024 rows (closed-form sine table extended by the caller). */
int cfd_load_tensor(cfd_handle h, const char* name, const float* data, size_t numel, int is_device);

/* After the last cfd_load_tensor: checks that every tensor is present, folds and re-lays-out the
 * weights on the device (float64 folding, then split-pair GEMM operands: two fp16 halves hi + lo per value). */
int cfd_finalize_weights(cfd_handle h);

/* Sinusoid rows of the timestep embedding for integer timesteps 0..n_rows-1
 * (get_timestep_embedding, convofusion/models/architectures/tools/embeddings.py:245-285, computed by
 * the caller exactly as the reference does: [cos | sin] halves, 512 floats per row), host float32. */
int cfd_set_timestep_table(cfd_handle h, const float* rows, int n_rows);

/* One memory of the conditioning tuple.  `data` [U][S][512] float32 (dev) holds the U DISTINCT
 * memories; `row_map` (dev int32 [Be], or NULL meaning U == Be, identity) says which one each
 * effective-batch row uses -- the 7-way guidance batch repeats each utterance's memory and one shared
 * unconditional memory (convofusion.py:909-929), which the memory-side projections exploit.
 * `key_padding_mask` [U][S] uint8 (dev, 1 = ignore key; nn.MultiheadAttention key_padding_mask,
 * cross_attention.py:587-626) or NULL. */
typedef struct {
  const float* data;
  const int32_t* row_map;
  const uint8_t* key_padding_mask;
  int U;
  int S;
} cfd_memory;

/* Replaces Denoiser.forward (denoiser.py:173-386).
 *   sample      dev [Be][L][128]
 *   timesteps   HOST int32, n_t == 1 (one timestep for every row, the sampler's case) or n_t == Be
 *   mem[5]      conditioning tuple
 *   out         dev [Be][L][128]   predicted noise
 *   att[5]      dev [Be][num_layers][L][S_j] attention probabilities (cross_attention.py:227-234), or
 *               NULL pointers to skip materialising them
 * Enqueued on `stream` (a hipStream_t, may be NULL = default stream); no host sync inside. */
int cfd_forward(cfd_handle h, const float* sample, int Be, int L, const int32_t* timesteps, int n_t,
                const cfd_memory mem[CFD_NUM_MEM], float* out, float* const att[CFD_NUM_MEM], void* stream);

/* The caller's promise for the NEXT cfd_forward on this handle: its memories (data, key-padding masks, row maps, shapes) are bit-identical
 * to those of the previous cfd_forward.  That call then reuses their timestep-independent projections (the folded keys / values of all
 * layers: most of a call's work when the reference's own Python loop calls the denoiser once per iteration with an un-de-duplicated 7 x B
 * conditioning batch, convofusion.py:499-513).  Ignored -- the projections are made -- whenever the handle cannot honour it: anything else
 * ran on the handle in between (a sampling run, a WEG evaluation, a forward with per-row timesteps), other shapes, weights reloaded. */
int cfd_forward_same_memories(cfd_handle h);

/* Replaces Convofusion._diffusion_reverse (convofusion/models/modeltype/convofusion.py:391-549) and
 * its in-painting copy diffusion_reverse_forecast (unbounded_synthesis.py:28-187), together with the
 * diffusers-0.14.0 scheduler calls inside them (set_timesteps / step / add_noise). */
typedef struct {
  int B;                      /* utterances */
  int L;                      /* latent tokens (16 in the product; must be even) */
  int G;                      /* guidance chunks: 7 (clf_guidance_drops + 1, convofusion.py:60) or 1 */
  float guidance_weight[8];   /* weight of chunk k >= 1 in  e_0 + sum_k w_k (e_k - e_0)
                                 reference: {-, 7.5, 7.5, 7.5, 7.5, 7.5, 0}  (convofusion.py:529-541) */
  int scheduler;              /* 0 = DDPMScheduler (fixed_small), 1 = DDIMScheduler */
  int num_train_timesteps;    /* 1000 */
  int num_inference_steps;    /* scheduler.set_timesteps(N) */
  int clip_sample;            /* configs/modules/scheduler.yaml:11 */
  float eta;                  /* DDIM only */
  int set_alpha_to_one;       /* DDIM only */
  int steps_offset;           /* DDIM only */
  const float* alphas_cumprod;/* HOST float32 [num_train_timesteps] (the scheduler's table) */
  const float* init_latents;  /* dev [B][L][128] N(0,1) draws (scaled by init_noise_sigma = 1), or NULL:
                                 drawn on the device, Philox stream 1 */
  const float* step_noise;    /* dev [iterations][B][L][128] or NULL: Philox stream 0 */
  uint64_t seed;              /* Philox key */
  uint32_t first_utterance;   /* global id of utterance 0 (shards draw independent sub-streams) */
  const float* preseq;        /* dev [B][preseq_len][128] previous-window latents to in-paint, or NULL */
  int preseq_len;
  cfd_memory mem[CFD_NUM_MEM];/* Be = G*B rows */
  int skip_zero_weight_chunks;/* != 0: trailing guidance chunks whose weight is exactly 0 are not evaluated.  The
                                 reference computes the full-conditioning chunk and multiplies it by
                                 guidance_scale * 0 (convofusion.py:538); its forward only feeds the per-step
                                 attention maps, which the fused loop does not keep.  Results are identical. */
  int dynamic_memory_mask;    /* bit j set: the CONTENTS of memory j may be rewritten by the caller between iterations of the
                                 run (the dyadic rollout's partner projection); its projections are then made in every
                                 iteration.  0 (the reference loop: memories are constants of a run, convofusion.py:391-549):
                                 the timestep-independent part of every memory's projections is computed once at
                                 cfd_sample_begin and the memories are not read again. */
  const int32_t* timesteps;   /* HOST int32 [num_timesteps]: the loop's timestep sequence `scheduler.timesteps` (convofusion.py:423),
                                 or NULL: (arange(N) * (T // N))[::-1] (+ steps_offset for DDIM), N = num_inference_steps.  The step
                                 formulas keep `prev_t = t - T // N` either way.  Needed for DDPM counts that do not divide T, where
                                 diffusers 0.14.0's table arange(0, T, T // N)[::-1] has MORE than N entries (unpinned, see
                                 convofusion_amd/scheduler.py); the run then has num_timesteps iterations. */
  int num_timesteps;
  float* att_ring[CFD_NUM_MEM];/* all NULL, or five dev buffers [iterations][B][num_layers][L][S_j] float32: the captured iteration
                                 stores the attention probabilities of the LAST guidance chunk (full conditioning) of iteration i into
                                 slot i -- the reference's per-iteration dict attention_matrices[t] = att_mats of the last chunk
                                 (convofusion.py:517-523, dumped as att_<t>.npy by base.py:243-259).  Small problems (the row-tile
                                 path) store them from their second cross-attention launch, all others from the fused
                                 cross-attention kernel's softmax (+ one small launch per iteration): 2 - 5 % of the run time.
                                 Needs skip_zero_weight_chunks == 0 and no dynamic memory (such a run has no fused cross-attention:
                                 cfd_sample_begin fails with CFD_E_SHAPE and the caller takes the maps with one cfd_forward per
                                 iteration).  The ring's size is the caller's business: iterations x B x layers x L x keys. */
  int operand_policy;         /* Operand format of the fused cross-attention's key / value tiles of the LONG memories (>= 128 padded keys: the
                                 audio memory) in THIS run (csrc/xattn_fused.hpp, OPF):
                                 0 = fp16 split pairs everywhere (3 MFMAs per product, ~2^-22 operand error: what cfd_forward always uses);
                                 bit 0 = their folded VALUES as single fp16 (the linear path of the attention; halves those tiles' L2 -> LDS
                                 traffic); bit 1 = their folded KEYS as single fp16 (the exponentiated path); bit 2 / bit 3 = the
                                 PROBABILITIES / the QUERIES of those products as one fp16 as well, i.e. plain fp16 attention against the long
                                 memories (1 MFMA per product instead of 3).  The shipped library implements the four bits together (any
                                 non-zero value = 15; the partial combinations were measured and are dominated: DESIGN.md section 2).
                                 Short memories (the text / speaker / activity memories: few keys, little averaging of the rounding) always
                                 keep pairs.  The reference is float32 throughout (cross_attention.py:593-652); which runs tolerate which bits
                                 is measured per scheduler in DESIGN.md section 2 -- convofusion_amd.sampler.OPERAND_POLICY holds the default
                                 per scheduler kind.  Ignored (pairs) on the row-tile path, with att_ring, and with a dynamic memory. */
} cfd_sample_args;

/* Opens a sampling run: builds the per-step coefficient and timestep-embedding tables, draws / copies
 * the initial latents and captures ONE loop iteration (replicate -> denoiser -> guidance -> scheduler
 * step) as a hipGraph on `stream`. */
int cfd_sample_begin(cfd_handle h, const cfd_sample_args* args, void* stream);
/* Replays the captured iteration `n` more times (asynchronously on the run's stream). */
int cfd_sample_steps(cfd_handle h, int n);
/* Number of iterations executed so far in the open run. */
int cfd_sample_position(cfd_handle h);

/* Dyadic reactive path (BASELINE.json configs[4]; the partner projection is the reference's TextAudioMotionFuser.latent_proj,
 * condfuser.py:22-27: Linear 128->128, GELU, Linear 128->out_dim, GELU): `n` lock-step iterations of two open sampling runs.
 * Each iteration enqueues, on ONE stream (side A's) and with no host synchronisation, the projection of side B's current latents
 * into side A's conditional speaker-memory rows `spk_a` [B][L][512], the projection of side A's latents into `spk_b`, side A's
 * captured iteration and side B's captured iteration.  Both runs must have been opened with the speaker memory declared dynamic
 * (cfd_sample_args.dynamic_memory_mask bit 0) and `spk_*` pointing into the memories they captured.  All pointers are device
 * pointers; `tmp` holds B * L * hidden floats.
 * Merged form (side_b == NULL), for two sides that share the denoiser's weights: ONE run of 2 B utterances -- side A's followed by
 * side B's -- whose conditional speaker rows [0, B) are `spk_a` and [B, 2 B) are `spk_b`; an iteration is the two projections and one
 * captured iteration of the double batch. */
typedef struct {
  const float* w1;  /* [hidden][latent_dim] */
  const float* b1;  /* [hidden] */
  const float* w2;  /* [out_dim][hidden] */
  const float* b2;  /* [out_dim] */
  int hidden, out_dim;
  float* spk_a;
  float* spk_b;
  float* tmp;
} cfd_dyadic_proj;
int cfd_dyadic_steps(cfd_handle side_a, cfd_handle side_b, const cfd_dyadic_proj* proj, int n);
/* Copies the current latents to `out` (dev [B][L][128]); with close != 0 also ends the run. */
int cfd_sample_read(cfd_handle h, float* out, int close);

/* Stand-alone scheduler ops on device tensors (diffusers 0.14.0 `scheduler.step(...).prev_sample` and
 * `add_noise`), for callers that drive their own loop (unbounded_synthesis.py:75,181). */
int cfd_scheduler_step(cfd_handle h, int scheduler, const float* alphas_cumprod, int num_train_timesteps,
                       int num_inference_steps, int t, int clip_sample, float eta, int set_alpha_to_one,
                       const float* model_output, const float* noise, float* sample_inout, size_t numel,
                       float* pred_original_sample /* dev [numel] or NULL: the (clipped) x0 estimate the step forms,
                       SchedulerOutput.pred_original_sample (read at convofusion.py:619) */,
                       void* stream);
int cfd_add_noise(cfd_handle h, const float* alphas_cumprod_host, int t, const float* original,
                  const float* noise, float* out, size_t numel, void* stream);

/* Conditioning producers on device tensors: out[r][n] = act(b[n] + sum_k x[r][k] W[n][k]), float32 -- one
 * nn.Linear (+ activation) of the small MLPs that make the memories: AudioConvEncoder.main / .out_net
 * (convofusion/models/architectures/audioenc.py:12-21,33-34: Linear, LeakyReLU(0.1), Linear, LeakyReLU(0.1), Linear;
 * dropout is the identity at inference) and TextAudioMotionFuser.latent_proj (condfuser.py:22-27: Linear, GELU,
 * Linear, GELU), which the dyadic path applies to the partner's current latents every step (BASELINE config 5).
 *   x dev [n_rows][K], W dev [N][K] (nn.Linear.weight), b dev [N] or NULL, out dev [n_rows][N]
 *   act: 0 none, 1 nn.GELU() (erf form), 2 nn.LeakyReLU(0.1) */
int cfd_linear_act(cfd_handle h, const float* x, long long n_rows, int K, const float* W, const float* b, int N, int act,
                   float* out, void* stream);

/* float32 building blocks of the small transformer that follows the loop -- ConvoFusionVae.decode
 * (convofusion/models/architectures/vae.py:268-372: two SkipTransformerDecoders of d_model 128, 2 heads, 5 pre-norm layers,
 * cross_attention.py:66-125,311-395) -- used by convofusion_amd/vae.py together with cfd_linear_act:
 *   cfd_layer_norm  nn.LayerNorm over the last dimension (D <= 2048)
 *   cfd_mha         the attention core of nn.MultiheadAttention on already projected q / k / v, sequence-major
 *                   [L][bs][E] rows, key_padding_mask dev uint8 [bs][Lk] (1 = ignore) or NULL; head_dim <= 64, Lk <= 1024
 *   cfd_add         x += y (residual connection)
 *   cfd_zero_rows   rows with keep[row] == 0 are zeroed (vae.py:358 `output[~mask.T] = 0`) */
int cfd_layer_norm(cfd_handle h, const float* x, long long rows, int D, const float* gamma, const float* beta, float eps,
                   float* out, void* stream);
int cfd_mha(cfd_handle h, const float* q, const float* k, const float* v, int Lq, int Lk, int bs, int E, int H,
            const uint8_t* key_padding_mask, float* out, void* stream);
int cfd_add(cfd_handle h, float* x, const float* y, size_t numel, void* stream);
int cfd_zero_rows(cfd_handle h, float* x, const uint8_t* keep, long long rows, int D, void* stream);

/* float32 pieces of word-excitation guidance (WEG): the attend-and-excite objective on the listener-text attention
 * maps and d(loss)/d(latents) through the denoiser -- what the reference gets from torch autograd over
 * Denoiser.forward (convofusion/models/modeltype/convofusion.py:437-496, iterative_refinement_step :298-388;
 * convofusion/models/tools/word_excitation_guidance.py:11-81).  convofusion_amd/weg.py strings them into the
 * forward-with-saved-activations and the hand-written backward pass; every transpose there is a strided view.
 *   cfd_mat             element (z1, z2, r, c) = p[z1*b1 + z2*b2 + r*rs + c*cs]
 *   cfd_gemm_f32        C(z; m, n) = alpha * sum_k A(z; m, k) B(z; k, n) + bias[n] (+ C when accumulate); nb1 x nb2 batch
 *   cfd_softmax         in-place softmax over [rows][Lk]; key_padding_mask [batch][Lk], batch = row / rows_per_batch
 *   cfd_softmax_bwd     dp <- p * ((dp + extra) - sum_k (dp + extra) p); extra (may be NULL) = gradient arriving at p directly
 *   cfd_layer_norm_bwd  nn.LayerNorm backward with respect to the input; accumulate != 0: dx += result
 *   cfd_ew              element-wise: 0 SiLU, 1 GELU, 2 a*SiLU'(b), 3 a*GELU'(b), 4 a + alpha*b (weg.update_latent),
 *                       5 a + b[r0*s0 + r1*s1 + d] (broadcast add over a [R0][R1][D] tensor), 6 TimeBlock modulate
 *                       a*(1 + b[r1][d]) + b[r1][D + d] (cross_attention.py:433-436), 7 its backward a*(1 + b[r1][d])
 *   cfd_weg_focus       aggregate_attentions + get_max_attention_at_indices (softmax over text[1:last), 3x3 Gaussian sigma 0.5
 *                       on the reflect-padded map, max over frames) + compute_attention_focus_loss, and the gradient with
 *                       respect to att: att / d_att dev [B][NL][L][S], tok_off dev int32 [B+1], tok_idx dev int32 (text
 *                       positions), kernel3 HOST {corner, edge, centre} of the normalised 3x3 kernel, workspace dev
 *                       >= B*(3*L*(last-1) + 3*nt_max) floats, losses dev [B], max_att dev [tok_off[B]]
 *   cfd_sample_write    overwrite the current latents of the open sampling run (the WEG update between two iterations)
 *   cfd_sample_inpaint  do the next iteration's in-painting overwrite of the first preseq_len tokens now (the captured
 *                       iteration then skips it): in the rollout the WEG update lands between the overwrite and the
 *                       replication (unbounded_synthesis.py:70-143); no-op without preseq */
typedef struct {
  const float* p;
  long long rs, cs, b1, b2;
} cfd_mat;
int cfd_gemm_f32(cfd_handle h, int M, int N, int K, int nb1, int nb2, const cfd_mat* A, const cfd_mat* B, const cfd_mat* C, const float* bias,
                 float alpha, int accumulate, void* stream);
int cfd_softmax(cfd_handle h, float* scores, long long rows, int Lk, const uint8_t* key_padding_mask, long long rows_per_batch, void* stream);
int cfd_softmax_bwd(cfd_handle h, const float* p, float* dp, const float* extra, long long rows, int Lk, void* stream);
int cfd_layer_norm_bwd(cfd_handle h, const float* x, const float* gamma, const float* dy, float* dx, long long rows, int D, float eps,
                       int accumulate, void* stream);
int cfd_ew(cfd_handle h, int op, const float* a, const float* b, float* out, size_t numel, int D, int R1, long long s0, long long s1, float alpha,
           void* stream);
int cfd_weg_focus(cfd_handle h, const float* att, int B, int NL, int L, int S, const int32_t* tok_off, const int32_t* tok_idx, int last, int nt_max,
                  const float kernel3[3], float* workspace, float* losses, float* max_att, float* d_att, void* stream);
/* One evaluation of the WEG objective on the text-only guidance chunk and d(loss)/d(latents), all launches enqueued from
 * C++: replaces  latents.requires_grad_(True); _, att = denoiser(latents, t, text_only_states, ...);
 * loss = compute_attention_focus_loss(get_max_attention_at_indices(aggregate_attentions(att[2]), ...));
 * torch.autograd.grad(loss, latents)   (convofusion.py:447-471,490-495; iterative_refinement_step :322-346,372-386).
 *   losses dev [B], max_att dev [tok_off[B]] (>= 1 float), grad dev [B][L][128]; loss_host (may be NULL): mean of losses,
 *   copied back after a stream synchronise (the loop branches on it: `loss > 1 - threshold`, `loss != 0`). */
typedef struct {
  int B, L;                      /* rows of the text-only chunk (the reference requires B == 1 with normalize_eot) */
  int timestep;
  const float* latents;          /* dev [B][L][128] */
  cfd_memory mem[CFD_NUM_MEM];   /* chunk 1 of the guidance batch: U == B, row_map NULL, masks as in cfd_forward */
  const int32_t* tok_off;        /* HOST [B + 1] offsets into tok_idx */
  const int32_t* tok_idx;        /* HOST focus token positions (text positions, BOS = 0), each in [1, last) */
  int last;                      /* text slice [1, last): eot index (normalize_eot) or S_text - 1 */
  float kernel3[3];              /* {corner, edge, centre} of GaussianSmoothing(1, 3, 0.5, dim=2).weight */
  int reuse_memory_side;         /* != 0: the memory CONTENTS (and masks) are those of the previous call, so their LayerNorms and
                                    key / value projections are taken from it (ignored when shapes / pointers differ).
                                    1: the timestep is the previous call's too (the refinement loop at one timestep,
                                       convofusion.py:322-346): its time embedding is reused as well.
                                    2: the timestep may differ (the guided sampling loop evaluates the objective once per
                                       iteration with the same conditioning, convofusion.py:437-471): small problems keep
                                       per-timestep tables for every timestep, built at the first such call (a few ms), and an
                                       evaluation selects its row; larger problems treat a new timestep like 0. */
} cfd_weg_args;
int cfd_weg_eval(cfd_handle h, const cfd_weg_args* args, float* losses, float* max_att, float* grad, float* loss_host, void* stream);
int cfd_sample_write(cfd_handle h, const float* latents);
int cfd_sample_inpaint(cfd_handle h);

/* Device N(0,1) draws of the product's counter-based stream (DESIGN.md "RNG"): out dev [B][per_utt]. */
int cfd_philox_normal(cfd_handle h, float* out, int B, int per_utt, uint64_t seed, uint32_t step,
                      uint32_t first_utterance, uint32_t stream_id, void* stream);

/* Measurement hook for bench.py: runs ONE denoiser forward of the currently configured problem eagerly
 * with every kernel class bracketed by HIP events on the launch stream; returns milliseconds per class
 * and the number of launches per class.  Classes: see CFD_PROF_* below. */
#define CFD_PROF_GEMM_TOKEN 0    /* token-side projections (QK, V^T, Wo, TimeBlocks, FFN, embed, proj) */
#define CFD_PROF_GEMM_MEM 1      /* memory-side K / V^T projections */
#define CFD_PROF_GEMM_ATTN 2     /* fused self-attention; score and P.V products of the three-launch cross-attention path */
#define CFD_PROF_ROWS 3          /* LayerNorm / AdaLN / softmax / memory prep */
#define CFD_PROF_OTHER 4
#define CFD_PROF_XATTN 5         /* fused cross-attention kernel (scores + softmax + P.V + residual of the five memories) */
#define CFD_PROF_NCLASS 6
int cfd_profile_forward(cfd_handle h, float ms[CFD_PROF_NCLASS], int launches[CFD_PROF_NCLASS]);

#ifdef __cplusplus
}
#endif
#endif
