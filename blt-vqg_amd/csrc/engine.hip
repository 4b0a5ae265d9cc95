// Train-step engine: the whole IQ forward / loss / backward / optimiser step as a fixed sequence of HIP kernel launches
// on one stream (static shapes: B, S_a, S_p, T are fixed per engine, so the sequence can be captured into a hipGraph).
// No autograd tape: the backward sequence is written by hand against the saved activations in the workspace arena.
//
// Mirrors (reference file:line): models/iq.py:82-114 (IQ.forward), models/encoder_cnn.py:30-35,
// models/encoder_transformer.py:22-37, models/decoder_transformer.py:22-41, models/transformer_layers.py:41-59,138-152,
// 205-221,260-282,326-364,400-408,486-532, models/mlp.py:49-56, train_iq.py:81-103 (losses), 259-261,372 (Adam, clip).
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "kernels.h"
#include "../../include/bltvqg_hip.h"

namespace {

struct PInfo {
    std::string name;
    int64_t off = 0, numel = 0;
    int dims[4] = {0, 0, 0, 0};
    int ndim = 0;
    int late = 0;
};

struct ConvSpec {
    std::string wname, bnname;
    int Cin, CinPad, Cout, K, stride, pad, Hi, Wi, Ho, Wo;
    void* wpacked = nullptr;
    float *scale = nullptr, *shift = nullptr;
    void* out = nullptr;   // raw conv output / in-place BN result: [B,Ho,Wo,Cout] (stem) or padded-pitch [B,Ho+1,Wo+1,Cout] (pp)
    bool pp = false;       // input AND output in the padded-pitch layout (every convolution after the stem / max-pool)
};

struct Layer {   // saved activations of one transformer layer
    void *xn1, *qkv, *ctx, *x1, *xn2, *h, *y2, *x2;
    float *m1, *r1, *m2, *r2;
    // decoder only
    void *q2, *kv2, *ctx2, *x1b, *xn3;
    float *m3, *r3;
    // {sum, sum of squares} per row of the inputs of this layer's LayerNorms, left by the GEMM epilogue that produced those rows, for
    // the Linear each LayerNorm is folded into (st1 of layer 0 = the embedding GEMM's rows: Stack::stat_in)
    float *st1 = nullptr, *st2 = nullptr, *st3 = nullptr;
    // backward: gradients that are OPERANDS of a (deferred) weight-gradient GEMM live in buffers of their own, never reused inside one
    // backward pass, so that those GEMMs can run on a side stream while the main stream walks on down the chain:
    // gY = d(FFN output before the residual), gF = d(FFN hidden), gQKV = d(q|k|v), dx1..3 = d(sub-layer outputs), and for the
    // decoder gQ = d(enc-dec query), gKV = d(enc-dec key|value)
    void *gY, *gF, *gQKV, *dx1, *dx2, *dx3, *gQ, *gKV;
};

struct Stack {
    std::string prefix;   // e.g. "answer_encoder.encoder"
    bool dec = false;
    int id = 0;           // dropout stream namespace
    int scr = 0;          // gradient scratch set used by this stack's backward
    int S = 0, M = 0;
    // only row 0 of every sample of this stack's output is read (the posterior encoder: encoder_transformer.py:35 takes
    // response_encoder_outputs[:, 0]): everything behind the TOP layer's attention core is row-wise, so it runs on those B rows only
    bool row0 = false;
    // ... its backward writes B strided rows of d(attention context) and d(sub-layer output) into buffers of their own whose other rows
    // are zero from the bind-time memset and are never written by anything else (the attention core and the first LayerNorm's backward
    // read ALL rows; the all-rows A/B form, debug key 23 bit 1, uses the ordinary buffers): no per-step memset.
    void *gA_top = nullptr, *dx1_top = nullptr;
    const int* key_ids = nullptr;
    void* x_in = nullptr;
    float* stat_in = nullptr;   // row statistics of x_in (the embedding GEMM's epilogue; rows_add_stat for the decoder's row 0)
    int stat_in_parts = 1;      // ... in that many partial-sum slots per row (the column tiles of the GEMM that wrote them)
    std::vector<Layer> layers;
    void* out = nullptr;
    float *mF = nullptr, *rF = nullptr;
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

}  // namespace

struct bltvqg_engine {
    bltvqg_config c;
    int dt, es;                 // dtype, element size
    int B, H, F, Z, E, L, NH, V, Sa, Sp, T;
    int Ma, Mp, Mt, Mtot, Epad, ldV, dh;
    // Padded model widths (bltvqg_config::head_dim_true): every H-wide tensor stores each head's head_dim_true real features in a slot of
    // dh = H / heads columns, the rest are zero pads (zero weights, zero gradients).  dh_true = real head width, H_true = heads * dh_true,
    // ln_pp / ln_pv = the LayerNorm kernels' pad pattern (0 / 0 when nothing is padded).
    int dh_true = 0, H_true = 0, ln_pp = 0, ln_pv = 0;
    int imgHp = 0, imgWp = 0;   // zero-bordered NHWC4 input image of the 7x7/2 stem
    float* ln_pool = nullptr;
    size_t ln_pool_floats = 0, ln_pool_used = 0;
    std::vector<LnRed> pending_ln;
    bool regions = false;       // BASELINE configs[4]: the image input is [B, num_regions, region_dim] precomputed features
    bool region_attn = false;   // ... pooled by region attention instead of the mean
    void *xr = nullptr, *Pr = nullptr, *dPr = nullptr;      // bf16/fp32 copy of the regions [B*R, D], projected regions [B*R, H] and their gradient
    float* ralpha = nullptr;    // attention weights [B, R]
    int FD = 512;               // width of the pooled feature the trainable head projects (512 = ResNet-18, region_dim in region mode)
    std::string fcw, fcb;       // the head's projection: encoder_cnn.cnn.fc.* (encoder_cnn.py:20) or encoder_cnn.region_proj.*
    std::vector<PInfo> tp, fp;
    std::unordered_map<std::string, int> ti, fi;      // (keyed access only; ~1 300 lookups per step on the host's enqueue path)
    int64_t tsize = 0, late_off = 0, fsize = 0, ws_bytes = 0;
    // bound memory
    float *train = nullptr, *grad = nullptr, *adam_m = nullptr, *adam_v = nullptr, *frozen = nullptr;
    char* ws = nullptr;
    bool bound = false, frozen_dirty = true, fwd_done = false;
    // The optimiser pass writes the plain bf16 weight shadow itself (misc.hip::adam_kernel); the next forward then only derives the
    // transposed copy from it.  That shortcut is valid only if nobody else wrote the fp32 parameters in between: the CALLER says so
    // (bltvqg_engine_trust_shadows: the fused step driver does, the autograd module whose parameters a torch optimiser updates does
    // not), and `shadow_gen` must equal the shared params_gen, which every optimiser step of ANY engine sharing these parameters and
    // every bltvqg_engine_invalidate_frozen (load_state) bumps.
    bool trust_shadows = false;
    long shadow_gen = -1;
    int phase2 = 0;
    uint64_t seed = 0;
    // Adam bias-correction counters of the always-trained / latent-phase-only regions.  They belong to the OPTIMISER STATE (the
    // moment buffers), not to an engine: engines of different batch shapes that share one set of parameter / moment buffers (the
    // ragged last batch of an epoch) share one counter object too (bltvqg_engine_share_optimizer_state).
    // params_gen: bumped by every write to the shared parameters.  opt_gen / opt_done_ev: the asynchronous optimiser update (below) is a
    // write to the SHARED buffers, so "an update is in flight" belongs here too: every engine sharing them orders itself behind the
    // latest update it has not yet waited for (engine B's forward must not run under engine A's in-flight Adam).
    struct AdamSteps { int main = 0, late = 0; long params_gen = 0; long opt_gen = 0; hipEvent_t opt_done_ev = nullptr, opt_stage1_ev = nullptr; };
    std::shared_ptr<AdamSteps> steps = std::make_shared<AdamSteps>();
    int last_bwd_phase2 = 0;
    // workspace buffers
    void* wshadow = nullptr;    // bf16 mirror of the flat trainable buffer (bf16 mode)
    void* wshadowT = nullptr;   // same offsets, every dgrad operand stored TRANSPOSED ([K,N], ld N): dX = dY W becomes an NT GEMM
    struct TEnt { int off, rows, cols, tile0; };
    std::vector<TEnt> tlist;    // matrices with a transposed shadow (fused q|k|v and k|v groups are one matrix)
    std::unordered_map<int64_t, int> trows;   // element offset -> rows of the transposed matrix registered there
    void* ttable = nullptr;     // device copy of tlist
    int ttiles = 0;
    void* wemb_pad = nullptr;   // padded shadow of embedding.1.weight when E % chunk != 0
    int ld_wemb = 0;
    float* timing = nullptr;
    int *ids_all, *pos_all, *tgt_shift, *tgt32, *ctx32, *post32;
    float* counters;
    float* stats;               // 8 floats
    void *img, *pool0, *feats;
    float *pooled, *featpre, *feats32, *dfeats32, *dfeatpre32;   // the CNN head (avg-pool -> fc -> BatchNorm1d) stays in fp32
    // ---- the frozen conv stack one batch ahead (bltvqg_engine_prefetch_images) --------------------------------------------------------
    // encoder_cnn.py:18-19 freezes the backbone: the conv stack of batch i+1 depends on nothing step i updates.  It runs on a stream of
    // its own (optionally on a CU partition of its own, bltvqg_engine_set_cu_masks) underneath step i and leaves the pooled [B,512]
    // feature in the OTHER slot of `pooled_buf`; forward(i+1) waits for that slot's event and starts at the trainable head.  BatchNorm2d
    // running statistics advance in batch order because the stack of every batch runs on that one in-order stream.
    float* pooled_buf[2] = {nullptr, nullptr};
    int pool_cur = 0;            // slot the current step's head (forward and weight gradient) reads
    int pf_head = 0, pf_n = 0;   // prefetched slots not yet consumed by a forward: pf_head is the next one, pf_n in {0, 1, 2}
    int prefetch_split = 10;     // leading stages of the conv stack (cnn_convs) a prefetch runs; the forward that consumes it runs the rest
    int pf_split[2] = {10, 10};
    bool pool_in_use = false;    // a forward consumed pool_cur and its backward has not been enqueued yet
    bool cnn_stream_used = false;
    int last_cnn_slot = 0;       // slot of the last stack enqueued on the conv stream (its event orders an inline stack behind it)
    hipStream_t cnn_stream = nullptr, chain_stream = nullptr;
    bool cnn_stream_owned = true;      // false: the caller's stream (bltvqg_engine_adopt_conv_stream), never destroyed here
    hipEvent_t cnn_in_ev = nullptr, cnn_done[2] = {nullptr, nullptr};
    bool have_masks = false;
    uint32_t cu_masks[3][8] = {};      // [0] chain (caller's stream + side[0]), [1] side[1] (weight gradients, optimiser), [2] conv stack
    bool cu_mask_on[3] = {false, false, false};
    int make_stream(hipStream_t* st, int which) {
        hipError_t rc = (have_masks && cu_mask_on[which]) ? hipExtStreamCreateWithCUMask(st, 8, cu_masks[which])
                                                          : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
        if (rc != hipSuccess) { blt_set_error("engine: stream creation failed (%s)", hipGetErrorString(rc)); return BLT_ERR_HIP; }
        return BLT_OK;
    }
    float *bn1_mean, *bn1_rstd;
    std::vector<ConvSpec> convs;
    float *stat_sum, *stat_sq;
    double* stat_tmp;
    void *emb_rows, *X_all;
    Stack enc, renc, dec;
    void *mlvp_h1, *mlvp_h2, *mlvp, *cat_in, *mlvq_h1, *mlvq_h2, *mlvq, *zlat, *zproj, *zc_in, *zlogit, *logits;
    void *r_in, *hrec, *recon;
    float* eps_dev;
    // gradient scratch
    void *sA[2], *sB[2], *sC[2];   // short-lived gradient scratch (consumed by the next launch): [0] main stream, [1] posterior-encoder stream
    float* acc_big;                // fp32 accumulator of the split-K vocabulary dgrad
    void *d_enc, *d_renc, *dX_all, *dE, *d_feats, *d_zproj, *d_recon, *dzl;
    void *g_b1, *g_b2, *g_b3, *g_b4, *g_cat, *g_mq;   // small [B, *] scratch
    void *g_rec1, *g_net[2][2];                      // [B, *] gradients that are operands of deferred weight-gradient GEMMs (never reused)
    void *g_rin, *g_zc;                              // d(reconstructor input), d(z_classifier input): produced on the branch stream
    float* acc_big2;                                 // split-K accumulator of the z_classifier dgrad (branch stream)
    // Gradient buckets (data-parallel exchange): contiguous ranges of the flat gradient buffer, listed in the order backward
    // COMPLETES them.  The weight gradients of a stack are flushed to the side stream in groups of whole layers of >= ~32 MB
    // (bucket_target_bytes), each flush closes one bucket and records its event right behind its last launch, so that every collective but
    // the last is enqueued while backward is still running and none is larger than a few layers.
    struct Bucket { int64_t off = 0, len = 0; int late = 0; hipEvent_t ev = nullptr; };
    std::vector<Bucket> buckets;                         // completion order
    std::vector<int> dec_flush, enc_flush, renc_flush;   // per layer l: bucket closed by the flush after layer l's backward, or -1
    int bk_dec_last = -1, bk_late0 = -1, bk_enc_last = -1, bk_renc_last = -1, bk_tail = -1;
    static constexpr int64_t BUCKET_TARGET_BYTES = 32ll << 20, BUCKET_TAIL_BYTES = 8ll << 20;
    // In-stack flushes exist for the data-parallel exchange (a bucket's all-reduce can start while backward is still running); on one GPU
    // they only take CUs from the chain (measured +0.15 ms per step on BASELINE configs[2]), so they are off until the step driver that owns
    // an exchange turns them on (bltvqg_engine_set_bucket_flush).  Off: one flush per stack, every bucket of the stack final with it.
    bool group_flush = false;
    int record_buckets(const std::vector<int>& plan, hipStream_t st) {
        for (int b : plan)
            if (b >= 0 && buckets[b].ev && hipEventRecord(buckets[b].ev, st) != hipSuccess) { blt_set_error("engine: bucket event record failed"); return BLT_ERR_HIP; }
        return BLT_OK;
    }
    // the 330 MB gradient memset leaves the critical path: forward() issues it on a side stream (behind the previous optimiser update,
    // the last reader of the gradients) and backward only waits for its event
    hipEvent_t grad_zero_ev = nullptr;
    bool grads_zeroed = false;
    int zero_grads_early(hipStream_t side_s) {
        if (!grad_zero_ev) return BLT_OK;
        if (hipMemsetAsync(grad, 0, sizeof(float) * (size_t)tsize, side_s) != hipSuccess || hipEventRecord(grad_zero_ev, side_s) != hipSuccess) {
            blt_set_error("engine_forward: early gradient memset failed");
            return BLT_ERR_HIP;
        }
        grads_zeroed = true;
        return BLT_OK;
    }
    // side streams: independent sub-graphs (CNN | posterior encoder | context encoder) run concurrently so that their small
    // launches (40-160 workgroups each) fill the 256 CUs together; fork/join with events (capturable into a hipGraph)
    // Only TWO side streams: HIP multiplexes streams onto 4 hardware queues by default, and a stream that shares a queue with the main
    // stream is silently serialised behind it (seen: the encoder stacks running after the CNN instead of beside it once the trainer's
    // communication streams were added).  side[0]: posterior encoder (forward and backward); side[1]: context encoder (forward), the
    // deferred weight-gradient GEMMs (backward) and the asynchronous optimiser update.
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t fj[16] = {};
    bool kv_hoisted = false;   // the decoder's encoder-side key/value projections were issued on the branch stream (forward_tail)
    // weight-gradient GEMMs of the transformer stacks are off the critical path of backward (nothing downstream reads dW): they are
    // collected while a stack's input-gradient chain is enqueued and then issued on a side stream, where they fill the CUs that the
    // chain's small latency-bound launches leave idle
    std::vector<GemmArgs> pending_wgrads;
    bool defer_wgrads = false;
    // Asynchronous optimiser (bltvqg_engine_optimizer_step_async): gradient norm + clip + Adam run on their own stream behind the
    // caller's stream (hence behind the gradient all-reduce the caller made that stream wait for), and the NEXT forward only makes
    // the consumers of trainable parameters wait for them: the frozen CNN (45 % of a step) starts at once, hiding the optimiser and
    // the tail of the all-reduce.  Every other entry point first orders itself behind the pending update (sync_opt).
    hipStream_t opt_stream = nullptr;
    hipEvent_t opt_fork = nullptr, opt_done = nullptr, opt_stage1 = nullptr;
    // Staged update: Adam runs over the parameters in the order the NEXT forward needs them — first the two encoder stacks, the shared
    // embedding and the CNN head (stage 1, its own event), then the decoder, vocabulary projection, reconstructor and the latent-phase heads
    // (stage 2).  The next forward's token / encoder streams wait for stage 1 only, so stage 2 (half of the 2.2 GB pass), the decoder's
    // transposed weight shadows and the gradient memset run underneath the encoder stacks instead of in front of them.
    int64_t dec_end = 0, renc_off = 0;            // [0, dec_end): stage-2 part of the main region; [renc_off, tsize): stage-1 part of the late region
    int t_dec_n = 0, t_main_n = 0, t_heads_n = 0; // tlist ranges: [0, t_dec_n) decoder side, [.., t_main_n) encoder + embedding, [.., t_heads_n) late heads, rest r_encoder
    long opt_seen = 0;           // the shared opt_gen this engine's caller stream is already ordered behind
    bool opt_is_pending() const { return opt_seen != steps->opt_gen; }
    void opt_mark_synced() { opt_seen = steps->opt_gen; }
    int sync_opt(hipStream_t s) {
        if (!opt_is_pending() || !steps->opt_done_ev) return BLT_OK;
        if (hipStreamWaitEvent(s, steps->opt_done_ev, 0) != hipSuccess) { blt_set_error("engine: optimiser wait failed"); return BLT_ERR_HIP; }
        return BLT_OK;
    }
    // ---- LayerNorm folded into the Linear that consumes it (round 4; bf16, unpadded widths) -------------------------------------------
    // Every LayerNorm inside the stacks feeds exactly one Linear (q|k|v, the cross-attention query, the first FFN layer):
    //   LN(x) W^T + b = rstd_m (x W'^T - mean_m s_n) + c_n,  W' = W diag(gamma), s_n = sum_k W'[n,k], c_n = sum_k beta[k] W[n,k] + b_n.
    // The GEMM reads the RAW rows x against W' (wshadowF, rebuilt with s / c by ONE launch per optimiser stage: blt_ln_fold_prepare) and
    // finishes the normalisation in its epilogue; the row sums come from the epilogue of the GEMM that produced x (GemmArgs::out_stat).
    // 42 of the 45 layernorm_fwd launches of a step disappear (the three stack-final ones stay); the normalised tensor the weight
    // gradient needs is written by the LayerNorm's BACKWARD launch instead (blt_layernorm_bwd xn_out).  debug key 25 = 1: unfolded (A/B).
    bool fold_ok = false;
    void* wshadowF = nullptr;
    float *fold_s = nullptr, *fold_c = nullptr;
    std::unordered_map<std::string, int> fold_srow;         // consumer weight -> first row in fold_s / fold_c
    std::vector<BltFoldEnt> fold_tab[2];          // [0] the two encoder stacks (stage-1 parameters of the optimiser), [1] the decoder
    int fold_nrows[2] = {0, 0};
    void* fold_tab_dev[2] = {nullptr, nullptr};
    int fold_rows_total = 0;
    float *stat_pool = nullptr, *stat_emb = nullptr;
    size_t stat_pool_floats = 0;
    // statistics slots per row: one per 64 columns of the row, whatever tile the producing GEMM takes (plain stores; the consumer adds them
    // in slot order: the sums do not depend on tile shapes or row counts)
    int stat_slots = 8;
    int stat_parts() const { return (H + 63) / 64; }      // partial sums per row: one per 64 columns
    bool fold_on() const { return fold_ok && blt_debug_get(25) != 1; }
    void add_fold(int which, const std::string& wname, int rows, const std::string& ln, const char* bias) {
        const PInfo& w = tpi(wname);
        BltFoldEnt e;
        e.w_off = w.off; e.g_off = tpi(ln + ".weight").off; e.b_off = tpi(ln + ".bias").off; e.bias_off = bias ? tpi(bias).off : -1;
        e.rows = rows; e.K = w.dims[1]; e.srow = fold_rows_total; e.row0 = fold_nrows[which];
        fold_srow[wname] = fold_rows_total;
        fold_rows_total += (rows + 7) / 8 * 8;
        fold_nrows[which] += rows;
        fold_tab[which].push_back(e);
    }
    void build_fold() {
        // (head-padded widths too — hidden 300 = 4 heads of 75 in 80-column slots: the pad columns of every activation, of gamma / beta and
        // of the consumers' weights are exact zeros, so the row sums, W' = W diag(gamma), s and c are those of the true width; only the divisor
        // is H_true.  debug key 25 = 2: folded LayerNorms for unpadded widths only — the round-4 state before this, A/B)
        fold_ok = dt == BLT_BF16 && H % 8 == 0 && (ln_pp == 0 || blt_debug_get(25) != 2);
        stat_slots = (H + 63) / 64;      // one slot per 64 columns of a row
        if (!fold_ok) return;
        const Stack* sts[3] = {&enc, &renc, &dec};
        for (const Stack* st : sts)
            for (int l = 0; l < L; ++l) {
                const std::string lp = st->prefix + (st->dec ? ".dec." : ".enc.") + std::to_string(l) + ".";
                const int w = st->dec ? 1 : 0;
                add_fold(w, lp + (st->dec ? "multi_head_attention_dec." : "multi_head_attention.") + "query_linear.weight", 3 * H,
                         lp + (st->dec ? "layer_norm_mha_dec" : "layer_norm_mha"), nullptr);
                if (st->dec) add_fold(w, lp + "multi_head_attention_enc_dec.query_linear.weight", H, lp + "layer_norm_mha_enc", nullptr);
                add_fold(w, lp + "positionwise_feed_forward.layers.0.weight", F, lp + "layer_norm_ffn", (lp + "positionwise_feed_forward.layers.0.bias").c_str());
            }
    }
    int fold_prepare(int parts, hipStream_t s) {
        if (!fold_on()) return BLT_OK;
        for (int w = 0; w < 2; ++w)
            if (((parts >> w) & 1) && !fold_tab[w].empty()) {
                const int rc = blt_ln_fold_prepare(train, wshadowF, fold_s, fold_c, fold_tab_dev[w], (int)fold_tab[w].size(), fold_nrows[w], s);
                if (rc) return rc;
            }
        return BLT_OK;
    }
    // the folded form of Y = LN(x) W^T (+ b): A = the raw rows, B = W', statistics in / mean, rstd out
    void set_fold(GemmArgs& g, const void* x, int ldx, const std::string& wname, const float* stat, int parts, float* m, float* r) {
        const PInfo& p = tpi(wname);
        g.A = x; g.lda = ldx;
        g.B = (const char*)wshadowF + p.off * 2; g.ldb = p.dims[1];
        g.bias = nullptr;
        const int sr = fold_srow.at(wname);
        g.stat_slots = stat_slots; g.fold_np = parts;
        g.fold_s = fold_s + sr; g.fold_c = fold_c + sr; g.fold_stat = stat; g.fold_mean = m; g.fold_rstd = r; g.fold_eps = 1e-5f;
        g.fold_n = ln_pp ? (float)(H / ln_pp * ln_pv) : (float)H;      // features the statistics cover (pad columns hold zeros)
    }
    void set_stat(GemmArgs& g, float* stat) { g.out_stat = stat; g.stat_slots = stat_slots; }
    bool use_streams = true;
    int causal_mode = 1;       // 1 = training mask (pad OR future -> -1e18), 2 = prefix decoding (future keys excluded)
    bool bn_train = true;      // false: BatchNorm layers use their running statistics (module.eval(), greedy decoding)
    int fork(hipStream_t from, hipStream_t to, hipEvent_t ev) {
        if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) { blt_set_error("engine: stream fork/join failed"); return BLT_ERR_HIP; }
        return BLT_OK;
    }
    // optional in-stream timing of the two dominant kernel families: one event pair per launch, on the stream of the launch.
    // class 0 = the convolution launches of the frozen ResNet-18 stack, class 1 = every Linear-layer GEMM (forward, input gradient,
    // weight gradient) of the transformer stacks / embedding / vocabulary projection / latent nets.  prof_mask bit c enables class c.
    // diagnostic: hipEvent stamps on the MAIN stream at the phase boundaries of a step (debug key 12 = 1): where the critical path spends
    // its time, measured without a profiler (rocprofv3 makes the host the bottleneck and distorts the overlap)
    hipEvent_t stamp_ev[12] = {};
    bool stamp_used[12] = {};
    void stamp(int i, hipStream_t s) {
        if (blt_debug_get(12) != 1 || i < 0 || i >= 12) return;
        if (!stamp_ev[i] && hipEventCreate(&stamp_ev[i]) != hipSuccess) return;
        (void)hipEventRecord(stamp_ev[i], s);
        stamp_used[i] = true;
    }
    int prof_mask = 0;
    struct ProfRec { hipEvent_t a = nullptr, b = nullptr; int cls = 0; int w = 1; double flops = 0.0; int sidx = 0; };
    // which of the step's streams a bracket sits on: 0 = the caller's, 1 / 2 = the engine's side streams, 3 = the conv look-ahead stream, 4 = other
    int stream_index(hipStream_t s) const { return s == side[0] ? 1 : s == side[1] ? 2 : (cnn_stream && s == cnn_stream) ? 3 : (s == opt_stream ? 2 : 0); }
    // class-1 launches that go through gemm() are bracketed every prof_stride-th time and counted prof_stride times (an event pair is a
    // ~5 us bubble on its stream: 210 pairs would stretch the one profiled step of bench.py by ~2.5 ms); grouped weight-gradient launches
    // (4 per step, 15 % of the family's flops each) and convolutions are always bracketed
    int prof_stride = 1, prof_gemm_count = 0;
    std::vector<ProfRec> prof;
    size_t prof_n = 0;
    int prof_begin(int cls, hipStream_t s) {
        if (!((prof_mask >> cls) & 1)) return -1;
        if (prof_n == prof.size()) {
            ProfRec r;
            if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
            prof.push_back(r);
        }
        prof[prof_n].cls = cls;
        prof[prof_n].w = 1;
        prof[prof_n].sidx = stream_index(s);
        (void)hipEventRecord(prof[prof_n].a, s);
        return (int)prof_n++;
    }
    void prof_end(int idx, hipStream_t s, double flops) {
        if (idx < 0) return;
        (void)hipEventRecord(prof[idx].b, s);
        prof[idx].flops = flops;
    }
    // every Linear-layer GEMM of the engine goes through here (class-1 bracket when enabled)
    int gemm(int dtype, const GemmArgs& g, hipStream_t s) {
        const bool sampled = (prof_mask & 2) && (prof_gemm_count++ % prof_stride) == 0;
        const int pi = sampled ? prof_begin(1, s) : -1;
        if (pi >= 0) prof[pi].w = prof_stride;
        const int rc = blt_gemm(dtype, g, s);
        prof_end(pi, s, 2.0 * (double)g.M * (double)g.N * (double)g.K);
        return rc;
    }

    // ---------------------------------------------------------------------------------------------
    void add_t(const std::string& n, int r, int cdim, int late) {
        PInfo p;
        p.name = n; p.late = late;
        p.ndim = cdim ? 2 : 1;
        p.dims[0] = r; p.dims[1] = cdim;
        p.numel = (int64_t)r * (cdim ? cdim : 1);
        p.off = tsize;
        tsize += (p.numel + 7) / 8 * 8;    // 16-byte aligned in the bf16 shadow as well
        ti[n] = (int)tp.size();
        tp.push_back(p);
    }
    void add_f(const std::string& n, int d0, int d1, int d2, int d3, int ndim) {
        PInfo p;
        p.name = n; p.ndim = ndim;
        p.dims[0] = d0; p.dims[1] = d1; p.dims[2] = d2; p.dims[3] = d3;
        p.numel = (int64_t)d0 * (ndim > 1 ? d1 : 1) * (ndim > 2 ? d2 : 1) * (ndim > 3 ? d3 : 1);
        p.off = fsize;
        fsize += (p.numel + 3) / 4 * 4;
        fi[n] = (int)fp.size();
        fp.push_back(p);
    }
    void add_mha(const std::string& pre, int late) {
        add_t(pre + "query_linear.weight", H, H, late);
        add_t(pre + "key_linear.weight", H, H, late);
        add_t(pre + "value_linear.weight", H, H, late);
        add_t(pre + "output_linear.weight", H, H, late);
    }
    void add_ffn(const std::string& pre, int late) {
        add_t(pre + "layers.0.weight", F, H, late);
        add_t(pre + "layers.0.bias", F, 0, late);
        add_t(pre + "layers.1.weight", H, F, late);
        add_t(pre + "layers.1.bias", H, 0, late);
    }
    void add_ln(const std::string& pre, int late) {
        add_t(pre + ".weight", H, 0, late);
        add_t(pre + ".bias", H, 0, late);
    }
    // closes the bucket [start, tsize) and returns its index in `tmp` (build order; build_params reorders into completion order)
    std::vector<Bucket> tmp_buckets;
    int64_t bucket_start = 0;
    int close_bucket(int late) {
        Bucket b; b.off = bucket_start; b.len = tsize - bucket_start; b.late = late;
        bucket_start = tsize;
        tmp_buckets.push_back(b);
        return (int)tmp_buckets.size() - 1;
    }
    // layers L-1 .. 0 of a stack in backward order: after which layers the collected weight gradients are flushed (>= target bytes pending)
    void add_enc_stack(const std::string& pre, int late, std::vector<int>& flush_after) {
        add_ln(pre + ".layer_norm", late);
        flush_after.assign(L, -1);
        for (int l = L - 1; l >= 0; --l) {
            const std::string lp = pre + ".enc." + std::to_string(l) + ".";
            add_mha(lp + "multi_head_attention.", late);
            add_ffn(lp + "positionwise_feed_forward.", late);
            add_ln(lp + "layer_norm_mha", late);
            add_ln(lp + "layer_norm_ffn", late);
            // ... and the last layers of an ENCODER stack one by one (when a layer is >= 8 MB): the two encoder chains end backward, so
            // whatever their last flush carries is exchanged after it — one layer each instead of a 36 MB group (exposed bytes 82 -> 34 MB
            // on BASELINE configs[2..3])
            const int64_t pend = (tsize - bucket_start) * 4;
            if (l > 0 && (pend >= BUCKET_TARGET_BYTES || (l <= 2 && pend >= BUCKET_TAIL_BYTES))) flush_after[l] = close_bucket(late);
        }
    }

    void build_params() {
        // ---- trainable, in the order backward completes them; buckets are closed at the weight-gradient flush points ----
        add_t("decoder.output.weight", V, H, 0);
        add_t("decoder.output.bias", V, 0, 0);
        // (the reconstructor's weight gradients are collected before the decoder chain starts: they leave with its first flush)
        add_t("image_reconstructor.layers.fc0.weight", F, H, 0);
        add_t("image_reconstructor.layers.fc0.bias", F, 0, 0);
        add_t("image_reconstructor.layers.fc1.weight", H, F, 0);
        add_t("image_reconstructor.layers.fc1.bias", H, 0, 0);
        add_ln("decoder.decoder.layer_norm", 0);
        dec_flush.assign(L, -1);
        for (int l = L - 1; l >= 0; --l) {
            const std::string lp = "decoder.decoder.dec." + std::to_string(l) + ".";
            add_mha(lp + "multi_head_attention_dec.", 0);
            add_mha(lp + "multi_head_attention_enc_dec.", 0);
            add_ffn(lp + "positionwise_feed_forward.", 0);
            add_ln(lp + "layer_norm_mha_dec", 0);
            add_ln(lp + "layer_norm_mha_enc", 0);
            add_ln(lp + "layer_norm_ffn", 0);
            if (l > 0 && (tsize - bucket_start) * 4 >= BUCKET_TARGET_BYTES) dec_flush[l] = close_bucket(0);
        }
        const int t_dec_last = close_bucket(0);
        dec_end = tsize;
        add_enc_stack("answer_encoder.encoder", 0, enc_flush);
        const int t_enc_last = close_bucket(0);
        add_t("embedding.1.weight", H, E, 0);
        add_t("embedding.1.bias", H, 0, 0);
        add_t("embedding.0.weight", V, E, 0);
        add_t(fcw, H, FD, 0);
        add_t(fcb, H, 0, 0);
        if (region_attn) add_t("encoder_cnn.region_attn.weight", 1, H, 0);
        add_t("encoder_cnn.bn.weight", H, 0, 0);
        add_t("encoder_cnn.bn.bias", H, 0, 0);
        const int t_tail = close_bucket(0);
        late_off = tsize;
        add_t("decoder.z_classifier.weight", V, H, 1);
        add_t("decoder.z_classifier.bias", V, 0, 1);
        add_t("latent_projection.weight", H, Z, 1);
        add_t("latent_projection.bias", H, 0, 1);
        const char* nets[2] = {"latent_layer.mean_logvar_posterior", "latent_layer.mean_logvar_prior"};
        for (int n = 0; n < 2; ++n) {
            const int din = (n == 0) ? 2 * H : H;
            add_t(std::string(nets[n]) + ".0.weight", 2 * Z, din, 1);
            add_t(std::string(nets[n]) + ".0.bias", 2 * Z, 0, 1);
            add_t(std::string(nets[n]) + ".3.weight", 2 * Z, 2 * Z, 1);
            add_t(std::string(nets[n]) + ".3.bias", 2 * Z, 0, 1);
            add_t(std::string(nets[n]) + ".6.weight", 2 * Z, 2 * Z, 1);
            add_t(std::string(nets[n]) + ".6.bias", 2 * Z, 0, 1);
        }
        const int t_late0 = close_bucket(1);
        renc_off = tsize;
        add_enc_stack("answer_encoder.r_encoder", 1, renc_flush);
        const int t_renc_last = close_bucket(1);
        // completion order: the decoder's groups, the latent-phase heads (final with the decoder's last flush), the two encoder stacks'
        // in-stack groups, their last groups, the tail (embedding, CNN head).  The POSTERIOR encoder's come first at each level: both chains
        // are ~60 latency-bound launches whatever their row count, and the posterior one starts earlier (it is forked right behind the latent
        // backward, the context one follows the CNN head's backward on the caller's stream) — profiles/r03_forced_dist_timeline.txt: posterior
        // groups final at 5.2 / 6.0 ms, context groups at 6.0 / 6.4 ms of that step
        {
            std::vector<int> order, remap(tmp_buckets.size(), -1);
            for (int l = L - 1; l >= 0; --l) if (dec_flush[l] >= 0) order.push_back(dec_flush[l]);
            order.push_back(t_dec_last); order.push_back(t_late0);
            for (int l = L - 1; l >= 0; --l) {
                if (renc_flush[l] >= 0) order.push_back(renc_flush[l]);
                if (enc_flush[l] >= 0) order.push_back(enc_flush[l]);
            }
            order.push_back(t_renc_last); order.push_back(t_enc_last); order.push_back(t_tail);
            for (size_t i = 0; i < order.size(); ++i) { remap[order[i]] = (int)i; buckets.push_back(tmp_buckets[order[i]]); }
            for (int l = 0; l < L; ++l) {
                if (dec_flush[l] >= 0) dec_flush[l] = remap[dec_flush[l]];
                if (enc_flush[l] >= 0) enc_flush[l] = remap[enc_flush[l]];
                if (renc_flush[l] >= 0) renc_flush[l] = remap[renc_flush[l]];
            }
            bk_dec_last = remap[t_dec_last]; bk_late0 = remap[t_late0]; bk_enc_last = remap[t_enc_last]; bk_renc_last = remap[t_renc_last];
            bk_tail = remap[t_tail];
        }

        // ---- transposed-shadow table: every weight that appears as the B operand of an input-gradient GEMM ----
        auto ends_with = [](const std::string& a, const char* suf) { const size_t n = strlen(suf); return a.size() >= n && a.compare(a.size() - n, n, suf) == 0; };
        for (const PInfo& p : tp) {
            if (p.ndim != 2 || p.name == "embedding.0.weight" || p.name == "encoder_cnn.region_attn.weight") continue;
            if (p.name == fcw && !region_attn) continue;      // the CNN / mean-pool head runs in fp32 on the master weights: no shadow
            int rows = p.dims[0];
            const bool encdec = p.name.find("multi_head_attention_enc_dec.") != std::string::npos;
            if (ends_with(p.name, "value_linear.weight")) continue;                       // part of a fused group
            if (ends_with(p.name, "key_linear.weight")) { if (!encdec) continue; rows = 2 * H; }      // enc-dec k|v
            if (ends_with(p.name, "query_linear.weight") && !encdec) rows = 3 * H;          // self-attention q|k|v
            // ld of the transposed operand must be 16-byte aligned; otherwise only the plain shadow is written (cols < 0 flags it)
            const bool tr = (rows % 8 == 0) && p.name != fcw;      // (the region projection's input has no gradient: plain shadow only)
            TEnt t; t.off = (int)p.off; t.rows = rows; t.cols = tr ? p.dims[1] : -p.dims[1]; t.tile0 = ttiles;
            ttiles += ((rows + 63) / 64) * ((p.dims[1] + 63) / 64);
            tlist.push_back(t);
            if (tr) trows[p.off] = rows;
        }
        for (const TEnt& t : tlist) {      // (tlist is in parameter order)
            if (t.off < dec_end) ++t_dec_n;
            if (t.off < late_off) ++t_main_n;
            if (t.off < renc_off) ++t_heads_n;
        }

        // ---- frozen backbone (torchvision resnet18 names, encoder_cnn.py:17) + running statistics ----
        auto add_bn = [&](const std::string& n, int C) {
            add_f(n + ".weight", C, 0, 0, 0, 1);
            add_f(n + ".bias", C, 0, 0, 0, 1);
            add_f(n + ".running_mean", C, 0, 0, 0, 1);
            add_f(n + ".running_var", C, 0, 0, 0, 1);
        };
        const std::string R = "encoder_cnn.cnn.";
        if (regions) {      // BASELINE configs[4]: precomputed region features, no backbone
            add_f("encoder_cnn.bn.running_mean", H, 0, 0, 0, 1);
            add_f("encoder_cnn.bn.running_var", H, 0, 0, 0, 1);
            return;
        }
        int hi = c.image_h, wi = c.image_w;
        auto add_conv = [&](const std::string& wn, const std::string& bn, int cin, int cout, int k, int s, int p, int h, int w) {
            add_f(wn, cout, cin, k, k, 4);
            add_bn(bn, cout);
            ConvSpec cs;
            cs.wname = wn; cs.bnname = bn; cs.Cin = cin; cs.CinPad = cin < 8 ? 4 : cin; cs.Cout = cout; cs.K = k;   // the stem (Cin = 3) runs on NHWC4
            cs.stride = s; cs.pad = p; cs.Hi = h; cs.Wi = w;
            cs.Ho = (h + 2 * p - k) / s + 1;
            cs.Wo = (w + 2 * p - k) / s + 1;
            convs.push_back(cs);
        };
        add_conv(R + "conv1.weight", R + "bn1", 3, 64, 7, 2, 3, hi, wi);
        hi = convs.back().Ho; wi = convs.back().Wo;
        hi = (hi + 2 - 3) / 2 + 1; wi = (wi + 2 - 3) / 2 + 1;   // maxpool 3x3/2 pad 1
        int cin = 64;
        const int couts[4] = {64, 128, 256, 512};
        for (int li = 0; li < 4; ++li)
            for (int b = 0; b < 2; ++b) {
                const int cout = couts[li];
                const int s = (li > 0 && b == 0) ? 2 : 1;
                const std::string bp = R + "layer" + std::to_string(li + 1) + "." + std::to_string(b) + ".";
                add_conv(bp + "conv1.weight", bp + "bn1", cin, cout, 3, s, 1, hi, wi);
                const int ho = convs.back().Ho, wo = convs.back().Wo;
                add_conv(bp + "conv2.weight", bp + "bn2", cout, cout, 3, 1, 1, ho, wo);
                if (s != 1 || cin != cout) add_conv(bp + "downsample.0.weight", bp + "downsample.1", cin, cout, 1, s, 0, hi, wi);
                hi = ho; wi = wo; cin = cout;
            }
        add_f("encoder_cnn.bn.running_mean", H, 0, 0, 0, 1);
        add_f("encoder_cnn.bn.running_var", H, 0, 0, 0, 1);
    }

    // ---------------------------------------------------------------------------------------------
    // workspace layout: called once with base = nullptr (sizes) and once at bind (pointers)
    // ---------------------------------------------------------------------------------------------
    int64_t layout(char* base) {
        int64_t off = 0;
        auto A = [&](int64_t bytes) -> void* {
            // integer arithmetic: the sizing pass runs with base = nullptr (pointer arithmetic on a null pointer is undefined behaviour)
            void* p = (void*)((uintptr_t)base + (uintptr_t)off);
            off += (bytes + 255) / 256 * 256;
            return p;
        };
        auto AT = [&](int64_t elems) -> void* { return A(elems * es); };
        auto AF = [&](int64_t n) -> float* { return (float*)A(n * 4); };
        auto AI = [&](int64_t n) -> int* { return (int*)A(n * 4); };
        wshadow = (dt == BLT_BF16) ? A(tsize * 2) : nullptr;
        wshadowT = (dt == BLT_BF16) ? A(tsize * 2) : nullptr;
        if (fold_ok) {
            wshadowF = A(tsize * 2);
            fold_s = AF(fold_rows_total); fold_c = AF(fold_rows_total);
            for (int w = 0; w < 2; ++w) fold_tab_dev[w] = A((int64_t)fold_tab[w].size() * sizeof(BltFoldEnt) + 64);
        }
        ttable = (dt == BLT_BF16) ? A((int64_t)tlist.size() * 16) : nullptr;
        const int ce = 16 / es;
        ld_wemb = round_up(E, ce);
        wemb_pad = (E % ce != 0) ? AT((int64_t)H * ld_wemb) : nullptr;
        timing = AF((int64_t)64 * H);
        ids_all = AI(Mtot); pos_all = AI(Mtot); tgt_shift = AI(Mt); tgt32 = AI(Mt); ctx32 = AI(Ma); post32 = AI(Mp);
        counters = AF(1 + B);
        stats = (float*)((uintptr_t)AF(12) + sizeof(float));         // stats[-1] = number of token ids outside [0, V) seen by prep_tokens (zeroed with the loss statistics)
        eps_dev = AF((int64_t)B * Z);
        // CNN
        img = regions ? nullptr : AT((int64_t)B * imgHp * imgWp * 4);
        int64_t max_stat = 0;
        for (auto& cs : convs) {
            cs.wpacked = AT((int64_t)cs.Cout * cs.K * (cs.Cin < 8 ? 8 : cs.K) * cs.CinPad);
            cs.scale = AF(cs.Cout); cs.shift = AF(cs.Cout);
            cs.pp = (&cs != &convs[0]);
            int64_t M = (int64_t)B * cs.Ho * cs.Wo;
            if (cs.pp) {      // guards of zero pixels in front of / behind the positions (the bind-time memset provides the zeros)
                M = blt_pp_pixels(B, cs.Ho, cs.Wo);
                char* base_ = (char*)AT((BLT_PP_GUARD_FRONT + M + BLT_PP_GUARD_TAIL) * cs.Cout);
                cs.out = (void*)((uintptr_t)base_ + (size_t)BLT_PP_GUARD_FRONT * cs.Cout * es);
            } else {
                cs.out = AT(M * cs.Cout);
            }
            const int64_t rows = 2 * ((M + 63) / 64);   // upper bound for either tile size
            if (rows * cs.Cout > max_stat) max_stat = rows * cs.Cout;
        }
        stat_sum = AF(max_stat); stat_sq = AF(max_stat);
        stat_tmp = (double*)A((int64_t)blt_bn_scratch_doubles(512) * 8);
        if (!regions) {
            const int ph = (convs[0].Ho + 2 - 3) / 2 + 1, pw = (convs[0].Wo + 2 - 3) / 2 + 1;
            char* base_ = (char*)AT((BLT_PP_GUARD_FRONT + blt_pp_pixels(B, ph, pw) + BLT_PP_GUARD_TAIL) * 64);
            pool0 = (void*)((uintptr_t)base_ + (size_t)BLT_PP_GUARD_FRONT * 64 * es);
        }
        if (region_attn) {
            const int64_t BR = (int64_t)B * c.num_regions;
            xr = AT(BR * FD); Pr = AT(BR * H); dPr = AT(BR * H);
            ralpha = AF(BR);
        }
        pooled_buf[0] = AF((int64_t)B * FD);
        pooled_buf[1] = regions ? pooled_buf[0] : AF((int64_t)B * FD);
        pooled = pooled_buf[pool_cur];
        featpre = AF((int64_t)B * H); feats32 = AF((int64_t)B * H); dfeats32 = AF((int64_t)B * H); dfeatpre32 = AF((int64_t)B * H);
        feats = AT((int64_t)B * H);
        bn1_mean = AF(H); bn1_rstd = AF(H);
        // embedding
        emb_rows = AT((int64_t)Mtot * Epad);
        X_all = AT((int64_t)Mtot * H);
        auto lay_stack = [&](Stack& s) {
            s.layers.resize(L);
            for (int l = 0; l < L; ++l) {
                Layer& y = s.layers[l];
                const int64_t M = s.M;
                y.xn1 = AT(M * H); y.qkv = AT(M * 3 * H); y.ctx = AT(M * H); y.x1 = AT(M * H); y.xn2 = AT(M * H);
                y.h = AT(M * F); y.y2 = AT(M * H); y.x2 = AT(M * H);
                y.m1 = AF(M); y.r1 = AF(M); y.m2 = AF(M); y.r2 = AF(M);
                y.gY = AT(M * H); y.gF = AT(M * F); y.gQKV = AT(M * 3 * H); y.dx1 = AT(M * H); y.dx2 = AT(M * H); y.dx3 = nullptr;
                y.gQ = nullptr; y.gKV = nullptr;
                if (s.dec) {
                    y.q2 = AT(M * H); y.kv2 = AT((int64_t)Ma * 2 * H); y.ctx2 = AT(M * H); y.x1b = AT(M * H); y.xn3 = AT(M * H);
                    y.m3 = AF(M); y.r3 = AF(M);
                    y.dx3 = AT(M * H); y.gQ = AT(M * H); y.gKV = AT((int64_t)Ma * 2 * H);
                }
            }
            s.out = AT((int64_t)s.M * H);
            s.mF = AF(s.M); s.rF = AF(s.M);
            if (s.row0) { s.gA_top = AT((int64_t)s.M * H); s.dx1_top = AT((int64_t)s.M * H); }
        };
        lay_stack(enc); lay_stack(renc); lay_stack(dec);
        if (fold_ok) {      // one contiguous pool: a single memset per forward
            const int64_t per_row = 2 * (int64_t)stat_slots;
            int64_t n = per_row * Mtot;
            for (Stack* st : {&enc, &renc, &dec}) n += (int64_t)L * (st->dec ? 3 : 2) * per_row * st->M;
            stat_pool_floats = (size_t)n;
            stat_pool = AF(n);
            float* q = stat_pool;
            stat_emb = q; q += per_row * Mtot;
            enc.stat_in = stat_emb; dec.stat_in = stat_emb + per_row * Ma; renc.stat_in = stat_emb + per_row * (Ma + Mt);
            for (Stack* st : {&enc, &renc, &dec})
                for (int l = 0; l < L; ++l) {
                    Layer& y = st->layers[l];
                    y.st1 = q; q += per_row * st->M;      // (layer 0 reads Stack::stat_in instead)
                    y.st2 = q; q += per_row * st->M;
                    if (st->dec) { y.st3 = q; q += per_row * st->M; }
                }
        }
        mlvp_h1 = AT((int64_t)B * 2 * Z); mlvp_h2 = AT((int64_t)B * 2 * Z); mlvp = AT((int64_t)B * 2 * Z);
        cat_in = AT((int64_t)B * 2 * H);
        mlvq_h1 = AT((int64_t)B * 2 * Z); mlvq_h2 = AT((int64_t)B * 2 * Z); mlvq = AT((int64_t)B * 2 * Z);
        zlat = AT((int64_t)B * Z); zproj = AT((int64_t)B * H); zc_in = AT((int64_t)B * H);
        zlogit = AT((int64_t)B * ldV);
        logits = AT((int64_t)Mt * ldV);
        r_in = AT((int64_t)B * H); hrec = AT((int64_t)B * F); recon = AT((int64_t)B * H);
        // gradient scratch
        const int64_t Mmax = (Mp > Mt ? Mp : Mt);
        for (int k = 0; k < 2; ++k) { sA[k] = AT(Mmax * H); sB[k] = AT(Mmax * H); sC[k] = AT(Mmax * H); }
        acc_big = AF(Mmax * H);
        d_enc = AT((int64_t)Ma * H); d_renc = AT((int64_t)Mp * H);
        dX_all = AT((int64_t)Mtot * H); dE = AT((int64_t)Mtot * Epad);
        d_feats = AT((int64_t)B * H); d_zproj = AT((int64_t)B * H); d_recon = AT((int64_t)B * H); dzl = AT((int64_t)B * ldV);
        const int64_t wide = (int64_t)B * (2 * Z > F ? 2 * Z : F);
        g_b1 = AT(wide); g_b2 = AT(wide); g_b3 = AT(wide); g_b4 = AT(wide); g_cat = AT((int64_t)B * 2 * H); g_mq = AT((int64_t)B * 2 * Z);
        g_rec1 = AT((int64_t)B * F);
        g_rin = AT((int64_t)B * H); g_zc = AT((int64_t)B * H);
        acc_big2 = AF((int64_t)B * H);
        for (int n = 0; n < 2; ++n) for (int k = 0; k < 2; ++k) g_net[n][k] = AT((int64_t)B * 2 * Z);
        // per-workgroup dgamma / dbeta partial sums of every LayerNorm backward of a step (ln_bwd)
        ln_pool_floats = ((size_t)(2 * L + 1) * (blt_layernorm_bwd_grid(Ma, H) + blt_layernorm_bwd_grid(Mp, H)) +
                          (size_t)(3 * L + 1) * blt_layernorm_bwd_grid(Mt, H)) * 2 * H;
        ln_pool = AF((int64_t)ln_pool_floats);
        wg_pool = (char*)A((int64_t)WG_SLOTS * WG_SLOT_BYTES);
        return off;
    }

    explicit bltvqg_engine(const bltvqg_config& cfg) : c(cfg) {
        dt = c.dtype; es = (dt == BLT_BF16) ? 2 : 4;
        B = c.batch; H = c.hidden_dim; F = c.pwffn_dim; Z = c.latent_dim; E = c.emb_dim; L = c.num_layers; NH = c.num_heads;
        V = c.vocab_size; Sa = c.len_context; Sp = c.len_posterior; T = c.len_target;
        Ma = B * Sa; Mp = B * Sp; Mt = B * T; Mtot = Ma + Mt + Mp;
        Epad = round_up(E, 32); ldV = round_up(V, 8); dh = H / NH;
        dh_true = (c.head_dim_true > 0 && c.head_dim_true < dh) ? c.head_dim_true : dh;
        H_true = NH * dh_true;
        if (dh_true != dh) { ln_pp = dh; ln_pv = dh_true; }
        {
            const int ho = (c.image_h + 6 - 7) / 2 + 1, wo = (c.image_w + 6 - 7) / 2 + 1;
            imgHp = c.image_h + 6 > 2 * (ho - 1) + 7 ? c.image_h + 6 : 2 * (ho - 1) + 7;
            imgWp = c.image_w + 6 > 2 * (wo - 1) + 8 ? c.image_w + 6 : 2 * (wo - 1) + 8;
            imgWp = (imgWp + 1) / 2 * 2;
        }
        regions = c.num_regions > 0;
        region_attn = regions && c.region_pool == 1;
        FD = regions ? c.region_dim : 512;
        fcw = regions ? "encoder_cnn.region_proj.weight" : "encoder_cnn.cnn.fc.weight";
        fcb = regions ? "encoder_cnn.region_proj.bias" : "encoder_cnn.cnn.fc.bias";
        build_params();
        enc.prefix = "answer_encoder.encoder"; enc.id = 0; enc.S = Sa; enc.M = Ma;
        renc.prefix = "answer_encoder.r_encoder"; renc.id = 1; renc.scr = 1; renc.S = Sp; renc.M = Mp; renc.row0 = true;
        dec.prefix = "decoder.decoder"; dec.id = 2; dec.S = T; dec.M = Mt; dec.dec = true;
        build_fold();
        ws_bytes = layout(nullptr);
    }

    // ---- parameter access ----------------------------------------------------------------------------
    const PInfo& tpi(const std::string& n) const { return tp[ti.at(n)]; }
    float* P(const std::string& n) const { return train + tpi(n).off; }
    float* G(const std::string& n) const { return grad + tpi(n).off; }
    float* FZ(const std::string& n) const { return frozen + fp[fi.at(n)].off; }
    // GEMM-operand view of a 2-D weight (shadow in bf16 mode)
    const void* W(const std::string& n, int* ld) const {
        const PInfo& p = tpi(n);
        if (n == "embedding.1.weight" && wemb_pad) { *ld = ld_wemb; return wemb_pad; }
        *ld = p.dims[1];
        if (dt == BLT_BF16) return (const char*)wshadow + p.off * 2;
        return train + p.off;
    }

    // ---- GEMM helpers ----------------------------------------------------------------------------------
    GemmArgs mk(const void* A_, int lda, int tA, const void* B_, int ldb, int tB, void* C_, int ldc, int M, int N, int K) {
        GemmArgs g;
        g.A = A_; g.lda = lda; g.transA = tA; g.B = B_; g.ldb = ldb; g.transB = tB; g.C = C_; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
        return g;
    }
    // Y[M,N] = X[M,K] W[N,K]^T (+bias)
    GemmArgs lin(const void* X, int ldx, const std::string& wname, const char* bias, void* Y, int ldy, int M) {
        int ldw;
        const void* w = W(wname, &ldw);
        const PInfo& p = tpi(wname);
        GemmArgs g = mk(X, ldx, 0, w, ldw, 0, Y, ldy, M, p.dims[0], p.dims[1]);
        if (bias) g.bias = P(bias);
        return g;
    }
    // transposed shadow of the [rows, K] matrix that starts at wname (rows > dims[0] for the fused q|k|v / k|v groups), or null
    const void* WT(const std::string& wname, int rows) const {
        if (dt != BLT_BF16) return nullptr;
        const PInfo& p = tpi(wname);
        auto it = trows.find(p.off);
        if (it == trows.end() || it->second != rows) return nullptr;
        return (const char*)wshadowT + p.off * 2;
    }
    // dX[M,K] = dY[M,N] W[N,K]: through the transposed shadow (k-contiguous B -> LDS-DMA kernel) when there is one
    GemmArgs dgrad_rows(const void* dY, int ldy, const std::string& wname, int rows, void* dX, int ldx, int M) {
        const PInfo& p = tpi(wname);
        if (const void* wt = WT(wname, rows)) return mk(dY, ldy, 0, wt, rows, 0, dX, ldx, M, p.dims[1], rows);
        int ldw;
        const void* w = W(wname, &ldw);
        return mk(dY, ldy, 0, w, ldw, 1, dX, ldx, M, p.dims[1], rows);
    }
    GemmArgs dgrad(const void* dY, int ldy, const std::string& wname, void* dX, int ldx, int M) {
        return dgrad_rows(dY, ldy, wname, tpi(wname).dims[0], dX, ldx, M);
    }
    // dW[N,K] += dY[M,N]^T X[M,K]  (fp32, into the flat gradient buffer) ; db[N] += colsum(dY)
    int wgrad(const void* dY, int ldy, const void* X, int ldx, const std::string& wname, const char* bias, int M, hipStream_t s) {
        const PInfo& p = tpi(wname);
        GemmArgs g = mk(dY, ldy, 1, X, ldx, 1, G(wname), p.dims[1], p.dims[0], p.dims[1], M);
        g.out_f32 = 1; g.split_k = 32;
        if (bias) g.a_rowsum = G(bias);       // bias gradient in the same launch
        return gemm(dt, g, s);
    }
    // same, but deferred to the weight-gradient stream when the caller's operands are stable (stack_bwd)
    int wgrad_later(const GemmArgs& g, hipStream_t s) {
        if (defer_wgrads) { pending_wgrads.push_back(g); return BLT_OK; }
        return gemm(dt, g, s);
    }
    int wgrad_later(const void* dY, int ldy, const void* X, int ldx, const std::string& wname, const char* bias, int M, hipStream_t s) {
        const PInfo& p = tpi(wname);
        GemmArgs g = mk(dY, ldy, 1, X, ldx, 1, G(wname), p.dims[1], p.dims[0], p.dims[1], M);
        g.out_f32 = 1; g.split_k = 32;
        if (bias) g.a_rowsum = G(bias);
        return wgrad_later(g, s);
    }
    // issue the collected weight-gradient GEMMs on `to`, ordered after everything enqueued on `from` so far.  bf16: ONE grouped launch
    // (gemm2.hip::wgrad_group_kernel) over a device-side problem table; the table of a flush point is the same every step (static
    // workspace pointers), so it is uploaded once and only re-uploaded when its content changes (phase switch).
    struct WgTable { std::vector<blt_wg_problem> probs; std::vector<int> wg0; int nwg = 0, bm = 128; bool valid = false; };
    std::vector<WgTable> wg_tables;
    int flush_idx = 0;
    char* wg_pool = nullptr;
    static constexpr int WG_SLOTS = 24, WG_SLOT_BYTES = 16384;
    // bucket >= 0: this flush makes that gradient bucket final — its event is recorded on `to` right behind the last launch
    int flush_wgrads(hipStream_t from, hipStream_t to, hipEvent_t ev, int bucket = -1, int bucket2 = -1) {
        const int rc0 = flush_wgrads_(from, to, ev);
        if (rc0) return rc0;
        const int bk[2] = {bucket, bucket2};
        for (int k = 0; k < 2; ++k)
            if (bk[k] >= 0 && bk[k] < (int)buckets.size() && buckets[bk[k]].ev && hipEventRecord(buckets[bk[k]].ev, to) != hipSuccess) {
                blt_set_error("flush_wgrads: bucket event record failed");
                return BLT_ERR_HIP;
            }
        return BLT_OK;
    }
    int flush_wgrads_(hipStream_t from, hipStream_t to, hipEvent_t ev) {
        if (pending_wgrads.empty() && pending_ln.empty()) return fork(from, to, ev);      // (the bucket event must still order behind `from`)
        int rc = fork(from, to, ev);
        bool grouped = dt == BLT_BF16 && !pending_wgrads.empty() && blt_debug_get(11) != 1 && flush_idx < WG_SLOTS && wg_pool != nullptr &&
                       pending_wgrads.size() * sizeof(blt_wg_problem) + (pending_wgrads.size() + 1) * 4 + 64 <= (size_t)WG_SLOT_BYTES;
        for (size_t i = 0; i < pending_wgrads.size() && grouped; ++i) grouped = blt_wgrad_group_ok(dt, pending_wgrads[i]);
        if (grouped && !rc) {
            if ((int)wg_tables.size() <= flush_idx) wg_tables.resize(flush_idx + 1);
            WgTable& t = wg_tables[flush_idx];
            std::vector<blt_wg_problem> probs;
            std::vector<int> wg0;
            int bm = 128;
            const int nwg = blt_wgrad_group_plan(pending_wgrads, probs, wg0, &bm);
            char* dev = wg_pool + (size_t)flush_idx * WG_SLOT_BYTES;
            const size_t pb = probs.size() * sizeof(blt_wg_problem), pb_al = (pb + 63) / 64 * 64;
            if (!t.valid || t.nwg != nwg || t.bm != bm || t.probs.size() != probs.size() || memcmp(t.probs.data(), probs.data(), pb) != 0) {
                t.probs = probs; t.wg0 = wg0; t.nwg = nwg; t.bm = bm; t.valid = true;
                if (hipMemcpyAsync(dev, t.probs.data(), pb, hipMemcpyHostToDevice, to) != hipSuccess ||
                    hipMemcpyAsync(dev + pb_al, t.wg0.data(), t.wg0.size() * 4, hipMemcpyHostToDevice, to) != hipSuccess) {
                    blt_set_error("flush_wgrads: table upload failed");
                    return BLT_ERR_HIP;
                }
            }
            double fl = 0.0;
            for (const GemmArgs& g : pending_wgrads) fl += 2.0 * (double)g.M * (double)g.N * (double)g.K;
            const int pi = (prof_mask & 2) ? prof_begin(1, to) : -1;
#ifdef BLT_ABLATE
            // debug key 14 (ablation build only): weight gradients are NOT computed — bounds what the side-stream launches cost the chain
            if (blt_debug_get(14) == 1) fl = 0.0;
            else
#endif
            rc = blt_wgrad_group_launch((const blt_wg_problem*)dev, (const int*)(dev + pb_al), (int)t.probs.size(), nwg, bm, to);
            prof_end(pi, to, fl);
            ++flush_idx;
        } else {
            for (size_t i = 0; i < pending_wgrads.size() && !rc; ++i) rc = gemm(dt, pending_wgrads[i], to);
        }
        pending_wgrads.clear();
        for (size_t i = 0; i < pending_ln.size() && !rc; i += BLT_LN_RED_MAX) {
            LnRedArgs a;
            a.n = (int)((pending_ln.size() - i < BLT_LN_RED_MAX) ? pending_ln.size() - i : BLT_LN_RED_MAX);
            for (int k = 0; k < a.n; ++k) a.e[k] = pending_ln[i + k];
            rc = blt_ln_param_reduce(a, to);
        }
        pending_ln.clear();
        return rc;
    }

    // LayerNorm backward; with deferred weight gradients its dgamma / dbeta go the same way: the launch on the dependent chain only
    // stores per-workgroup partial sums, one reduce launch per flush adds them on the side stream
    int ln_bwd(const void* dy, const void* x, const std::string& ln, const float* mean, const float* rstd, const void* dres, void* dx, long M,
               hipStream_t s, const void* maskY = nullptr, float mask_scale = 1.f, void* out2 = nullptr, long ld = 0, void* xn_out = nullptr) {
        float* part = nullptr;
        const int grid = blt_layernorm_bwd_grid(M, H);
        if (defer_wgrads && blt_debug_get(7) != 4 && ln_pool && ln_pool_used + (size_t)grid * 2 * H <= ln_pool_floats) {
            part = ln_pool + ln_pool_used;
            ln_pool_used += (size_t)grid * 2 * H;
            LnRed e; e.part = part; e.dgamma = G(ln + ".weight"); e.dbeta = G(ln + ".bias"); e.nblocks = grid; e.cols = H;
            pending_ln.push_back(e);
        }
        return blt_layernorm_bwd(dt, dy, x, P(ln + ".weight"), mean, rstd, dres, dx, G(ln + ".weight"), G(ln + ".bias"), M, H, s, maskY, mask_scale,
                                 out2, part, ln_pp, ln_pv, ld, xn_out ? P(ln + ".bias") : nullptr, xn_out);
    }

    // dX = dY W with a vocabulary-sized contraction (K = V): few output tiles and a long K loop, so the K range is split over
    // workgroups into an fp32 scratch (atomics) and cast back; falls through to the plain kernel in fp32 mode / small V
    int dgrad_bigk(const void* dY, int ldy, const std::string& wname, void* dX, int ldx, int M, hipStream_t s, float* acc_buf = nullptr,
                   size_t acc_floats = 0) {
        if (dt != BLT_BF16 || tpi(wname).dims[0] < 2048) return gemm(dt, dgrad(dY, ldy, wname, dX, ldx, M), s);
        // enough rows to fill the chip with 64x64 tiles (the decoder's output projection: 40 x 4 tiles, 125 K-tiles each on a deep ring):
        // the k-contiguous form through the transposed shadow needs no fp32 scratch, memset or cast
        if (M >= 1024 && WT(wname, tpi(wname).dims[0]) != nullptr) return gemm(dt, dgrad(dY, ldy, wname, dX, ldx, M), s);
        GemmArgs g;
        {   // n-contiguous B (the plain shadow): the split-K path lives in the register-staged kernel
            int ldw;
            const void* w = W(wname, &ldw);
            const PInfo& p = tpi(wname);
            g = mk(dY, ldy, 0, w, ldw, 1, dX, ldx, M, p.dims[1], p.dims[0]);
        }
        float* acc = acc_buf ? acc_buf : acc_big;
        if ((size_t)M * g.N > (acc_buf ? acc_floats : (size_t)(Mp > Mt ? Mp : Mt) * H)) return gemm(dt, g, s);
        if (hipMemsetAsync(acc, 0, (size_t)M * g.N * 4, s) != hipSuccess) { blt_set_error("dgrad_bigk: memset failed"); return BLT_ERR_HIP; }
        g.C = acc; g.ldc = g.N; g.out_f32 = 1; g.split_k = 16;
        { const int rc_ = gemm(dt, g, s); if (rc_) return rc_; }
        return blt_cast_rows(BLT_F32, acc, g.N, dt, dX, ldx, M, g.N, s);
    }

    uint32_t sid(int stack, int layer, int site) const { return (uint32_t)(stack * 1000 + layer * 10 + site); }

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

    // ---------------------------------------------------------------------------------------------
    // forward pieces
    // ---------------------------------------------------------------------------------------------
    int attn_fwd(const void* q, int ldq, const void* k, const void* v, int ldkv, void* o, const int* key_ids, int Tq, int Tk,
                 int causal, uint32_t stream_id, hipStream_t s) {
        AttnArgs a;
        a.Q = q; a.ldq = ldq; a.K = k; a.V = v; a.ldk = ldkv; a.ldv = ldkv; a.O = o; a.ldo = H; a.key_ids = key_ids;
        a.B = B; a.heads = NH; a.Tq = Tq; a.Tk = Tk; a.d = dh; a.causal = causal; a.scale = 1.f / sqrtf((float)dh_true);
        a.drop_p = c.attention_dropout; a.seed = seed; a.stream_id = stream_id;
        return blt_attn_fwd(dt, a, s);
    }
    // Rows of the row-wise part of layer l (everything behind the attention core): all M rows at stride H, or — top layer of a stack
    // whose output is read at row 0 of every sample only — B rows at stride S * H (row b of the view = row b * S of the tensor).  Nothing
    // that was computed for the other rows was ever read: the reference computes them and drops them (encoder_transformer.py:24,35).
    // debug key 23 bit 1: all rows (A/B, and the bit-identity test).
    struct Rows { int M; int ldH; int ldF; bool sub; };
    Rows rows_of(const Stack& st, int l) const {
        if (st.row0 && l == L - 1 && !(blt_debug_get(23) & 2)) return Rows{B, st.S * H, st.S * F, true};
        return Rows{st.M, H, F, false};
    }
    int ln_fwd(const void* x, const std::string& ln, void* out, float* m, float* r, const Rows& rw, hipStream_t s) {
        return blt_layernorm_fwd(dt, x, P(ln + ".weight"), P(ln + ".bias"), out, m, r, rw.M, H, 1e-5f, s, ln_pp, ln_pv, rw.ldH);
    }
    // attention core over all rows, then its output Linear (+ the sub-layer's residual in the epilogue) over the rows of `rw`
    int attn_out_fwd(const void* q, int ldq, const void* k, const void* v, int ldkv, void* ctx, const std::string& wname, const void* resid, void* out,
                     const int* key_ids, int Tq, int Tk, int causal, uint32_t stream_id, const Rows& rw, float* out_stat, hipStream_t s) {
        RC(attn_fwd(q, ldq, k, v, ldkv, ctx, key_ids, Tq, Tk, causal, stream_id, s));
        GemmArgs g = lin(ctx, rw.ldH, wname, nullptr, out, rw.ldH, rw.M);
        g.R = resid; g.ldr = rw.ldH;
        if (out_stat) set_stat(g, out_stat);      // row sums of `out` for the LayerNorm that reads it (folded into ITS consumer)
        return gemm(dt, g, s);
    }

    int attn_bwd(const void* q, int ldq, const void* k, const void* v, int ldkv, const void* dO, void* dq, int lddq, void* dk,
                 void* dv, int lddkv, const int* key_ids, int Tq, int Tk, int causal, uint32_t stream_id, hipStream_t s) {
        AttnArgs a;
        a.Q = q; a.ldq = ldq; a.K = k; a.V = v; a.ldk = ldkv; a.ldv = ldkv; a.key_ids = key_ids;
        a.B = B; a.heads = NH; a.Tq = Tq; a.Tk = Tk; a.d = dh; a.causal = causal; a.scale = 1.f / sqrtf((float)dh_true);
        a.drop_p = c.attention_dropout; a.seed = seed; a.stream_id = stream_id;
        a.dO = dO; a.lddo = H; a.dQ = dq; a.lddq = lddq; a.dK = dk; a.dV = dv; a.lddk = lddkv; a.lddv = lddkv;
        return blt_attn_bwd(dt, a, s);
    }

    // x2 = xres + dropout(relu(W2 dropout(relu(W1 xn + b1)) + b2)), xn = `ln`(xres): computed by a launch of its own (m / r / xn written
    // there) or — fold_on() — inside the first Linear from the raw rows and their sums `stat` (m / r written by that GEMM).  out_stat: row
    // sums of x2 for the next layer's first LayerNorm.
    int ffn_fwd(const std::string& fp_, const std::string& ln, void* xn, float* m, float* r, const float* stat, const void* xres, Layer& y,
                const Rows& rw, int stack, int l, float* out_stat, hipStream_t s) {
        GemmArgs g;
        if (fold_on()) {
            g = mk(nullptr, 0, 0, nullptr, 0, 0, y.h, rw.ldF, rw.M, F, H);
            set_fold(g, xres, rw.ldH, fp_ + "layers.0.weight", stat, stat_parts(), m, r);      // (xres = an attention output projection's result)
        } else {
            RC(ln_fwd(xres, ln, xn, m, r, rw, s));
            g = lin(xn, rw.ldH, fp_ + "layers.0.weight", (fp_ + "layers.0.bias").c_str(), y.h, rw.ldF, rw.M);
        }
        g.relu = 1; g.drop_p = c.relu_dropout; g.seed = seed; g.stream_id = sid(stack, l, 1);
        RC(gemm(dt, g, s));
        g = lin(y.h, rw.ldF, fp_ + "layers.1.weight", (fp_ + "layers.1.bias").c_str(), y.x2, rw.ldH, rw.M);
        g.relu = 1; g.drop_p = c.relu_dropout; g.seed = seed; g.stream_id = sid(stack, l, 2);
        g.C2 = y.y2; g.ldc2 = rw.ldH; g.R = xres; g.ldr = rw.ldH;
        if (out_stat) set_stat(g, out_stat);
        return gemm(dt, g, s);
    }

    // encoder-side key/value projections of every decoder layer ([2H,H] fused operand): they depend on encoder_outputs only, so they
    // are issued ahead of the decoder stack on the branch stream `sb`; the stack waits for fj[13] before its first cross-attention
    int dec_kv_fwd(hipStream_t sb) {
        for (int l = 0; l < L; ++l) {
            const std::string a2 = dec.prefix + ".dec." + std::to_string(l) + ".multi_head_attention_enc_dec.";
            int ldw;
            const void* w = W(a2 + "key_linear.weight", &ldw);
            RC(gemm(dt, mk(enc.out, H, 0, w, ldw, 0, dec.layers[l].kv2, 2 * H, Ma, 2 * H, H), sb));
        }
        if (hipEventRecord(fj[13], sb) != hipSuccess) { blt_set_error("engine_forward: event record failed"); return BLT_ERR_HIP; }
        kv_hoisted = true;
        return BLT_OK;
    }

    int stack_fwd(Stack& st, const void* enc_out, const int* src_ids, hipStream_t s) {
        const int M = st.M, S = st.S;
        const void* x = st.x_in;
        auto lname = [&](int l) { return st.prefix + (st.dec ? ".dec." : ".enc.") + std::to_string(l) + "."; };
        const Rows all{M, H, F, false};
        const bool fold = fold_on();
        for (int l = 0; l < L; ++l) {
            Layer& y = st.layers[l];
            const std::string lp = lname(l);
            const std::string a1 = lp + (st.dec ? "multi_head_attention_dec." : "multi_head_attention.");
            const std::string ln1 = lp + (st.dec ? "layer_norm_mha_dec" : "layer_norm_mha");
            const Rows rw = rows_of(st, l);
            // fused QKV projection: query/key/value weights are adjacent in the flat buffer -> one [3H,H] operand
            if (fold) {
                GemmArgs g = mk(nullptr, 0, 0, nullptr, 0, 0, y.qkv, 3 * H, M, 3 * H, H);
                // (x = the embedding GEMM's rows, or the second FFN Linear's result of the layer below — never the row-0-only top layer's)
                set_fold(g, x, H, a1 + "query_linear.weight", l == 0 ? st.stat_in : y.st1, l == 0 ? st.stat_in_parts : stat_parts(), y.m1, y.r1);
                RC(gemm(dt, g, s));
            } else {
                RC(ln_fwd(x, ln1, y.xn1, y.m1, y.r1, all, s));
                int ldw;
                const void* w = W(a1 + "query_linear.weight", &ldw);
                RC(gemm(dt, mk(y.xn1, H, 0, w, ldw, 0, y.qkv, 3 * H, M, 3 * H, H), s));
            }
            const std::string ln2 = lp + (st.dec ? "layer_norm_mha_enc" : "layer_norm_ffn");
            RC(attn_out_fwd(y.qkv, 3 * H, (char*)y.qkv + (size_t)H * es, (char*)y.qkv + (size_t)2 * H * es, 3 * H, y.ctx, a1 + "output_linear.weight",
                            x, y.x1, st.key_ids, S, S, st.dec ? causal_mode : 0, sid(st.id, l, 0), rw, fold ? y.st2 : nullptr, s));
            float* next_stat = (fold && l + 1 < L) ? st.layers[l + 1].st1 : nullptr;
            if (st.dec) {
                const std::string a2 = lp + "multi_head_attention_enc_dec.";
                if (fold) {
                    GemmArgs g = mk(nullptr, 0, 0, nullptr, 0, 0, y.q2, H, M, H, H);
                    set_fold(g, y.x1, H, a2 + "query_linear.weight", y.st2, stat_parts(), y.m2, y.r2);
                    RC(gemm(dt, g, s));
                } else {
                    RC(ln_fwd(y.x1, ln2, y.xn2, y.m2, y.r2, rw, s));
                    RC(gemm(dt, lin(y.xn2, H, a2 + "query_linear.weight", nullptr, y.q2, H, M), s));
                }
                if (!kv_hoisted) {
                    int ldw;
                    const void* w = W(a2 + "key_linear.weight", &ldw);
                    RC(gemm(dt, mk(enc_out, H, 0, w, ldw, 0, y.kv2, 2 * H, Ma, 2 * H, H), s));
                } else if (l == 0 && hipStreamWaitEvent(s, fj[13], 0) != hipSuccess) {
                    blt_set_error("engine_forward: stream wait failed");
                    return BLT_ERR_HIP;
                }
                RC(attn_out_fwd(y.q2, H, y.kv2, (char*)y.kv2 + (size_t)H * es, 2 * H, y.ctx2, a2 + "output_linear.weight", y.x1, y.x1b, src_ids, S, Sa,
                                0, sid(st.id, l, 3), rw, fold ? y.st3 : nullptr, s));
                RC(ffn_fwd(lp + "positionwise_feed_forward.", lp + "layer_norm_ffn", y.xn3, y.m3, y.r3, y.st3, y.x1b, y, rw, st.id, l, next_stat, s));
            } else {
                RC(ffn_fwd(lp + "positionwise_feed_forward.", ln2, y.xn2, y.m2, y.r2, y.st2, y.x1, y, rw, st.id, l, next_stat, s));
            }
            x = y.x2;
        }
        // the stack's final LayerNorm (transformer_layers.py:150,219) over the rows the top layer produced: a launch of its own (its
        // consumers are not one Linear: the row-0 injections, the cross-attention K/V projections of six layers, the vocabulary projection)
        return ln_fwd(x, st.prefix + ".layer_norm", st.out, st.mF, st.rF, rows_of(st, L - 1), s);
    }

    // ---- incremental greedy decoding (round 4).  The reference re-decodes the whole prefix for every new token
    // (iq.py:134-141: decoder.inference_forward on ys[:, :t+1]); round 2's form did the same with one decoder pass over all T rows per step.
    // Position t's activations depend on the tokens 0..t only and those never change once written, so step t computes ROW t of every
    // sample (B strided rows of the [B*T, .] tensors: row b*T + t, no gather) and attends over the q|k|v rows the earlier steps left in the
    // layer buffers — they ARE the key / value cache.  Self-attention: one query against keys 0..t (pad keys replaced by -1e18 as in the
    // full pass; "future keys do not exist" is the key count).  Cross-attention: the hoisted encoder-side key / value projections.
    int attn_row(const void* q, int ldq, const void* k, const void* v, int ldkv, int k_rows, void* o, const int* key_ids, int Tk, hipStream_t s) {
        AttnArgs a;
        a.Q = q; a.ldq = ldq; a.q_rows = T; a.K = k; a.V = v; a.ldk = ldkv; a.ldv = ldkv; a.k_rows = k_rows; a.O = o; a.ldo = H; a.key_ids = key_ids;
        a.B = B; a.heads = NH; a.Tq = 1; a.Tk = Tk; a.d = dh; a.causal = 0; a.scale = 1.f / sqrtf((float)dh_true);
        return blt_attn_fwd(dt, a, s);
    }
    int dec_step_fwd(int t, const int* src_ids, hipStream_t s) {
        Stack& st = dec;
        const Rows rw{B, T * H, T * F, true};
        const size_t oH = (size_t)t * H * es, oF = (size_t)t * F * es, o3 = (size_t)t * 3 * H * es;
        auto at = [](void* p, size_t off) { return (void*)((char*)p + off); };
        const bool fold = fold_on();
        const void* x = (const char*)st.x_in + oH;
        for (int l = 0; l < L; ++l) {
            Layer& y = st.layers[l];
            const std::string lp = st.prefix + ".dec." + std::to_string(l) + ".";
            const std::string a1 = lp + "multi_head_attention_dec.", a2 = lp + "multi_head_attention_enc_dec.";
            // q | k | v of row t (k and v join the cache)
            if (fold) {
                GemmArgs g = mk(nullptr, 0, 0, nullptr, 0, 0, at(y.qkv, o3), T * 3 * H, B, 3 * H, H);
                if (l == 0) {      // the embedding GEMM's statistics are indexed by the row of the whole tensor
                    set_fold(g, x, T * H, a1 + "query_linear.weight", st.stat_in + 2 * (size_t)stat_slots * t, st.stat_in_parts, y.m1, y.r1);
                    g.fold_sstride = T * stat_slots;
                } else {
                    set_fold(g, x, T * H, a1 + "query_linear.weight", y.st1, stat_parts(), y.m1, y.r1);
                }
                RC(gemm(dt, g, s));
            } else {
                RC(ln_fwd(x, lp + "layer_norm_mha_dec", at(y.xn1, oH), y.m1, y.r1, rw, s));
                int ldw;
                const void* w = W(a1 + "query_linear.weight", &ldw);
                RC(gemm(dt, mk(at(y.xn1, oH), T * H, 0, w, ldw, 0, at(y.qkv, o3), T * 3 * H, B, 3 * H, H), s));
            }
            RC(attn_row(at(y.qkv, o3), 3 * H, (char*)y.qkv + (size_t)H * es, (char*)y.qkv + (size_t)2 * H * es, 3 * H, T, at(y.ctx, oH), st.key_ids, t + 1, s));
            {
                GemmArgs g = lin(at(y.ctx, oH), T * H, a1 + "output_linear.weight", nullptr, at(y.x1, oH), T * H, B);
                g.R = x; g.ldr = T * H;
                if (fold) set_stat(g, y.st2);
                RC(gemm(dt, g, s));
            }
            // cross-attention: query of row t against the encoder-side keys / values (hoisted: dec_kv_fwd)
            if (fold) {
                GemmArgs g = mk(nullptr, 0, 0, nullptr, 0, 0, at(y.q2, oH), T * H, B, H, H);
                set_fold(g, at(y.x1, oH), T * H, a2 + "query_linear.weight", y.st2, stat_parts(), y.m2, y.r2);
                RC(gemm(dt, g, s));
            } else {
                RC(ln_fwd(at(y.x1, oH), lp + "layer_norm_mha_enc", at(y.xn2, oH), y.m2, y.r2, rw, s));
                RC(gemm(dt, lin(at(y.xn2, oH), T * H, a2 + "query_linear.weight", nullptr, at(y.q2, oH), T * H, B), s));
            }
            RC(attn_row(at(y.q2, oH), H, y.kv2, (char*)y.kv2 + (size_t)H * es, 2 * H, 0, at(y.ctx2, oH), src_ids, Sa, s));
            {
                GemmArgs g = lin(at(y.ctx2, oH), T * H, a2 + "output_linear.weight", nullptr, at(y.x1b, oH), T * H, B);
                g.R = at(y.x1, oH); g.ldr = T * H;
                if (fold) set_stat(g, y.st3);
                RC(gemm(dt, g, s));
            }
            Layer yt = y;      // the FFN's buffers at row t
            yt.h = at(y.h, oF); yt.x2 = at(y.x2, oH); yt.y2 = at(y.y2, oH);
            float* next_stat = (fold && l + 1 < L) ? st.layers[l + 1].st1 : nullptr;
            RC(ffn_fwd(lp + "positionwise_feed_forward.", lp + "layer_norm_ffn", at(y.xn3, oH), y.m3, y.r3, y.st3, at(y.x1b, oH), yt, rw, st.id, l, next_stat, s));
            x = yt.x2;
        }
        return ln_fwd(x, st.prefix + ".layer_norm", at(st.out, oH), st.mF, st.rF, rw, s);
    }

    // the LDS-patch kernel can take the previous convolution's raw output and apply its BatchNorm + ReLU on the staged patch
    bool conv_is_direct(const ConvSpec& cs) const {
        return cs.pp && dt == BLT_BF16 && cs.K == 3 && cs.stride == 1 && cs.pad == 1 && cs.Cin % 64 == 0 && cs.Cout % 64 == 0 && cs.Wo <= 62;
    }
    // stem + max-pool as one launch writing pooling-window extrema (conv_pp.hip); debug key 18 = 1: the two-pass form
    bool stem_pooled() const {
        return !regions && blt_debug_get(18) != 1 && blt_conv_stem_pool_ok(dt, c.image_h, c.image_w, imgHp, imgWp, convs.empty() ? 0 : convs[0].Cout);
    }
    int conv_fwd(ConvSpec& cs, const void* x, hipStream_t s, const ConvSpec* in_bn = nullptr) {
        const bool stem = cs.Cin < 8;
        // 3x3 stride-1 convolutions on padded-pitch bf16 activations: the LDS-patch kernel (conv_pp.hip); everything else (stem,
        // stride-2, 1x1, fp32 mode) is an implicit GEMM, reading / writing the padded-pitch layout through the generalised loader
        const bool direct = conv_is_direct(cs);
        if (in_bn != nullptr && !direct) { blt_set_error("conv_fwd: fused input BatchNorm needs the LDS-patch kernel"); return BLT_ERR_STATE; }
        GemmArgs g;
        g.A = x; g.B = cs.wpacked; g.C = cs.out;
        g.M = B * cs.Ho * cs.Wo; g.N = cs.Cout; g.K = stem ? 224 : cs.K * cs.K * cs.CinPad;
        g.lda = cs.CinPad; g.ldb = g.K; g.ldc = cs.Cout;
        g.is_conv = stem ? 2 : 1;
        g.cg.Hi = stem ? imgHp : cs.Hi; g.cg.Wi = stem ? imgWp : cs.Wi;
        g.cg.Cin = cs.CinPad; g.cg.cin_log2 = ilog2(cs.CinPad); g.cg.Ho = cs.Ho; g.cg.Wo = cs.Wo;
        g.cg.KH = cs.K; g.cg.KW = stem ? 8 : cs.K; g.cg.stride = cs.stride; g.cg.pad = stem ? 0 : cs.pad;
        if (cs.pp) {
            g.cg.in_rows = cs.Hi + 1; g.cg.in_pitch = cs.Wi + 1;
            g.cg.Hov = cs.Ho; g.cg.Wov = cs.Wo; g.cg.Ho = cs.Ho + 1; g.cg.Wo = cs.Wo + 1;
            g.M = B * g.cg.Ho * g.cg.Wo;
        }
        if (bn_train) { g.stat_sum = stat_sum; g.stat_sq = stat_sq; }
        const int pi = prof_begin(0, s);
        const bool direct_stem = stem && blt_conv_stem_direct_ok(dt, c.image_h, c.image_w, imgHp, imgWp, cs.Cout);
        const bool pooled_stem = stem && stem_pooled();
        if (pooled_stem) RC(blt_conv_stem_pool(x, cs.wpacked, FZ(cs.bnname + ".weight"), pool0, B, c.image_h, c.image_w, imgHp, imgWp, g.stat_sum, g.stat_sq, s));
        else if (direct) RC(blt_conv3x3_pp(x, cs.wpacked, cs.out, B, cs.Ho, cs.Wo, cs.Cin, cs.Cout, g.stat_sum, g.stat_sq, s, in_bn ? in_bn->scale : nullptr,
                                      in_bn ? in_bn->shift : nullptr));
        else if (direct_stem) RC(blt_conv_stem_direct(x, cs.wpacked, cs.out, B, c.image_h, c.image_w, imgHp, imgWp, g.stat_sum, g.stat_sq, s));
        else RC(blt_gemm(dt, g, s));
        prof_end(pi, s, 2.0 * (double)B * cs.Ho * cs.Wo * (double)cs.Cout * (double)(cs.K * cs.K * cs.Cin));   // algorithmic: real pixels, unpadded Cin
        if (!bn_train)
            return blt_bn_eval_scale(FZ(cs.bnname + ".weight"), FZ(cs.bnname + ".bias"), FZ(cs.bnname + ".running_mean"), FZ(cs.bnname + ".running_var"),
                                     1e-5f, cs.scale, cs.shift, cs.Cout, s);
        const int nparts = pooled_stem ? blt_conv_stem_pool_stat_rows(B, c.image_h, c.image_w)
                           : direct ? blt_conv3x3_pp_stat_rows_for(B, cs.Ho, cs.Wo, cs.Cin, cs.Cout)
                           : direct_stem ? blt_conv_stem_direct_stat_rows(B, c.image_h, c.image_w) : blt_gemm_stat_rows(g, dt);
        return blt_bn_finalize(stat_sum, stat_sq, nparts, cs.Cout, (long)B * cs.Ho * cs.Wo, FZ(cs.bnname + ".weight"), FZ(cs.bnname + ".bias"), 1e-5f,
                               0.1f, FZ(cs.bnname + ".running_mean"), FZ(cs.bnname + ".running_var"), cs.scale, cs.shift, nullptr,
                               nullptr, stat_tmp, s);
    }
    // BatchNorm apply (+ residual, ReLU) in place on a convolution's padded-pitch output: pad positions become zeros
    int bn_act(ConvSpec& cs, const void* res, const ConvSpec* res_bn, int relu, hipStream_t s) {
        return blt_bn_apply_pp(dt, cs.out, cs.scale, cs.shift, res, res_bn ? res_bn->scale : nullptr, res_bn ? res_bn->shift : nullptr, cs.out, B,
                               cs.Ho, cs.Wo, cs.Cout, relu, s);
    }

    // Image feature.  Image mode: EncoderCNN.forward (encoder_cnn.py:30-35).  Region mode (BASELINE configs[4], SURVEY A2': no
    // reference symbol): mean_r(Linear(D->H)(x_r)) -> the same BatchNorm1d; the mean commutes with the Linear, so the regions are
    // pooled first and the projection runs on [B, D] (36x fewer flops, same function).
    int cnn_fwd(const float* images, hipStream_t s) {
        if (regions) {
            RC(sync_opt(s));
            if (region_attn) {
                // p_r = Linear(D -> H)(x_r) for every region (one GEMM over [B*R, D]), then attention pooling -> the BatchNorm1d input
                const int BR = B * c.num_regions;
                RC(blt_cast_rows(BLT_F32, images, FD, dt, xr, FD, BR, FD, s));
                RC(gemm(dt, lin(xr, FD, fcw, fcb.c_str(), Pr, H, BR), s));
                RC(blt_region_attn_fwd(dt, Pr, P("encoder_cnn.region_attn.weight"), featpre, ralpha, B, c.num_regions, H, s));
                return cnn_head_fwd(s);
            }
            RC(blt_avgpool(BLT_F32, images, pooled, B, c.num_regions, FD, 1, s));
            return cnn_head_fwd(s);
        }
        // a batch whose conv stack ran ahead (prefetch_images): wait for its pooled feature, start at the trainable head
        if (pf_n > 0) {
            BLT_REQUIRE(images == nullptr, "engine_forward: a prefetched batch is pending, pass images = NULL (its images were given to prefetch_images)");
            const int slot = pf_head;
            if (hipStreamWaitEvent(s, cnn_done[slot], 0) != hipSuccess) { blt_set_error("engine_forward: prefetch wait failed"); return BLT_ERR_HIP; }
            pool_cur = slot; pooled = pooled_buf[slot];
            pf_head ^= 1; --pf_n;
            pool_in_use = true;
            // the stages the prefetch left (bltvqg_engine_set_prefetch_split) run here, behind the prefetched ones
            if (pf_split[slot] < CNN_STAGES) RC(cnn_convs(nullptr, s, pooled, pf_split[slot], CNN_STAGES));
            RC(sync_opt(s));
            return cnn_head_fwd(s);
        }
        // inline: the conv buffers are shared with the prefetch stream, so order behind whatever it still runs
        if (cnn_stream_used && hipStreamWaitEvent(s, cnn_done[last_cnn_slot], 0) != hipSuccess) { blt_set_error("engine_forward: conv-stream wait failed"); return BLT_ERR_HIP; }
        RC(cnn_convs(images, s, pooled));
        // Head in fp32 (exact-fp32 MFMA on the fp32 master weights): BatchNorm1d removes the common mode of the pooled feature
        // across the batch, so bf16 rounding of these tiny [B,512]/[B,H] tensors would be amplified into the image feature.
        RC(sync_opt(s));      // the head (fc + BatchNorm1d) is trainable: behind a pending asynchronous optimiser update
        return cnn_head_fwd(s);
    }

    // the frozen part of EncoderCNN.forward (encoder_cnn.py:33 minus the fc) as 10 stages: [0] image pack -> stem + BatchNorm + ReLU +
    // max-pool, [1..8] the eight BasicBlocks, [9] global average pool into `pooled_out` [B,512] fp32.  Runs stages [from, to).  Touches no
    // trainable parameter.
    static constexpr int CNN_STAGES = 10;
    int cnn_convs(const float* images, hipStream_t s, float* pooled_out, int from = 0, int to = CNN_STAGES) {
        if (frozen_dirty) {
            for (auto& cs : convs)
                RC(blt_conv_pack_w(dt, FZ(cs.wname), cs.wpacked, cs.Cout, cs.Cin, cs.K, cs.K, cs.CinPad, cs.Cin < 8 ? 8 : cs.K, s));
            frozen_dirty = false;
        }
#ifdef BLT_ABLATE
        if (blt_debug_get(15) == 1) return BLT_OK;      // timing ablation only: the conv stack is skipped, `pooled` keeps the last step's features
#endif
        size_t ci = 0;
        ConvSpec& c1 = convs[ci++];
        if (from <= 0 && to > 0) {
            if (images) RC(blt_img_pack(dt, images, img, B, 3, c.image_h, c.image_w, 4, 3, 3, imgHp, imgWp, s));      // NULL: the caller filled `img`
            RC(conv_fwd(c1, img, s));
            if (stem_pooled()) RC(blt_bn_apply_pp(dt, pool0, c1.scale, c1.shift, nullptr, nullptr, nullptr, pool0, B, c1.Ho / 2, c1.Wo / 2, 64, 1, s));
            else RC(blt_bn_relu_maxpool_pp(dt, c1.out, c1.scale, c1.shift, pool0, B, c1.Ho, c1.Wo, 64, s));
        }
        const void* x = pool0;
        int cin = 64;
        const int couts[4] = {64, 128, 256, 512};
        for (int li = 0; li < 4; ++li)
            for (int b = 0; b < 2; ++b) {
                const int stage = 1 + li * 2 + b;
                const int cout = couts[li];
                const int st = (li > 0 && b == 0) ? 2 : 1;
                ConvSpec& ca = convs[ci++];
                ConvSpec& cb = convs[ci++];
                ConvSpec* cd = (st != 1 || cin != cout) ? &convs[ci++] : nullptr;
                if (stage >= from && stage < to) {
                    RC(conv_fwd(ca, x, s));
                    // bn1 + ReLU: inside conv2's patch staging when conv2 is the LDS-patch kernel (debug key 17 = 1: the separate pass)
                    const bool fuse_bn1 = conv_is_direct(cb) && blt_debug_get(17) != 1;
                    if (!fuse_bn1) RC(bn_act(ca, nullptr, nullptr, 1, s));
                    RC(conv_fwd(cb, ca.out, s, fuse_bn1 ? &ca : nullptr));
                    const void* res = x;
                    if (cd) {      // downsample branch: its BatchNorm is applied inside the block's final BN + add + ReLU pass
                        RC(conv_fwd(*cd, x, s));
                        res = cd->out;
                    }
                    RC(bn_act(cb, res, cd, 1, s));
                }
                x = cb.out;
                cin = cout;
            }
        if (to >= CNN_STAGES && from < CNN_STAGES) return blt_avgpool_pp(dt, x, pooled_out, B, convs.back().Ho, convs.back().Wo, 512, 1, s);
        return BLT_OK;
    }

    // Enqueues the conv stack of the NEXT batch on the engine's conv stream, ordered behind everything enqueued on `sin` so far (the
    // producer of `images`; and, when the slot it fills is the one the previous step read, that step's backward).
    int prefetch_images(const float* images, hipStream_t sin) {
        BLT_REQUIRE(bound && !regions, "engine_prefetch_images: engine not bound / region mode has no conv stack");
        BLT_REQUIRE(bn_train, "engine_prefetch_images: train-mode BatchNorm only (eval engines run the stack inline)");
        BLT_REQUIRE(pf_n < 2, "engine_prefetch_images: two prefetched batches are already pending");
        // a PARTIAL stack leaves its intermediate activation in the (single-buffered) conv workspace until the forward that finishes it
        BLT_REQUIRE(pf_n == 0 || (prefetch_split >= CNN_STAGES && pf_split[pf_head] >= CNN_STAGES),
                    "engine_prefetch_images: with a partial prefetch (set_prefetch_split) only one batch may be pending");
        // the slot to fill: behind the pending one if there is one, else the one the current step does not read
        const int slot = pf_n > 0 ? (pf_head ^ 1) : (pool_cur ^ 1);
        BLT_REQUIRE(!(slot == pool_cur && pool_in_use), "engine_prefetch_images: the slot to fill is still read by the current step: enqueue its backward first");
        if (!cnn_stream) RC(make_stream(&cnn_stream, 2));
        RC(fork(sin, cnn_stream, cnn_in_ev));
        RC(cnn_convs(images, cnn_stream, pooled_buf[slot], 0, prefetch_split));
        pf_split[slot] = prefetch_split;
        if (hipEventRecord(cnn_done[slot], cnn_stream) != hipSuccess) { blt_set_error("engine_prefetch_images: event record failed"); return BLT_ERR_HIP; }
        if (pf_n == 0) pf_head = slot;
        ++pf_n;
        cnn_stream_used = true; last_cnn_slot = slot;
        return BLT_OK;
    }

    int cnn_head_fwd(hipStream_t s) {
        if (!region_attn) {      // (region attention: featpre already holds the pooled projection)
            const PInfo& pw = tpi(fcw);
            GemmArgs g = mk(pooled, FD, 0, train + pw.off, FD, 0, featpre, H, B, H, FD);
            g.bias = P(fcb);
            RC(gemm(BLT_F32, g, s));
        }
        if (bn_train) {
            RC(blt_bn1d_fwd(BLT_F32, featpre, P("encoder_cnn.bn.weight"), P("encoder_cnn.bn.bias"), feats32, bn1_mean, bn1_rstd,
                            FZ("encoder_cnn.bn.running_mean"), FZ("encoder_cnn.bn.running_var"), B, H, 1e-5f, 0.01f, s));
        } else {      // bn1_mean / bn1_rstd double as the eval-mode scale / shift
            RC(blt_bn_eval_scale(P("encoder_cnn.bn.weight"), P("encoder_cnn.bn.bias"), FZ("encoder_cnn.bn.running_mean"),
                                 FZ("encoder_cnn.bn.running_var"), 1e-5f, bn1_mean, bn1_rstd, H, s));
            RC(blt_bn_apply(BLT_F32, featpre, bn1_mean, bn1_rstd, nullptr, feats32, B, H, 0, s));
        }
        return blt_cast_rows(BLT_F32, feats32, H, dt, feats, H, B, H, s);
    }

    int mlp3_fwd(const std::string& net, const void* x, int din, void* h1, void* h2, void* out, hipStream_t s) {
        GemmArgs g = lin(x, din, net + ".0.weight", (net + ".0.bias").c_str(), h1, 2 * Z, B);
        g.relu = 1;   // the ReLU that opens the next Sequential stage is applied to the stored activation
        RC(gemm(dt, g, s));
        g = lin(h1, 2 * Z, net + ".3.weight", (net + ".3.bias").c_str(), h2, 2 * Z, B);
        g.relu = 1;
        RC(gemm(dt, g, s));
        return gemm(dt, lin(h2, 2 * Z, net + ".6.weight", (net + ".6.bias").c_str(), out, 2 * Z, B), s);
    }
    // backward of mlp3: dout [B,2Z] -> parameter grads, dx [B,din] (written, or accumulated into dx if acc)
    int mlp3_bwd(const std::string& net, int which, const void* x, int din, const void* h1, const void* h2, const void* dout, void* dx, int lddx,
                 int acc, hipStream_t s) {
        void *gh2 = g_net[which][0], *gh1 = g_net[which][1];      // operands of the deferred weight gradients: buffers of their own
        RC(wgrad_later(dout, 2 * Z, h2, 2 * Z, net + ".6.weight", (net + ".6.bias").c_str(), B, s));
        GemmArgs g = dgrad(dout, 2 * Z, net + ".6.weight", gh2, 2 * Z, B);
        g.maskY = h2; g.ldm = 2 * Z; g.mask_scale = 1.f;
        RC(gemm(dt, g, s));
        RC(wgrad_later(gh2, 2 * Z, h1, 2 * Z, net + ".3.weight", (net + ".3.bias").c_str(), B, s));
        g = dgrad(gh2, 2 * Z, net + ".3.weight", gh1, 2 * Z, B);
        g.maskY = h1; g.ldm = 2 * Z; g.mask_scale = 1.f;
        RC(gemm(dt, g, s));
        RC(wgrad_later(gh1, 2 * Z, x, din, net + ".0.weight", (net + ".0.bias").c_str(), B, s));
        g = dgrad(gh1, 2 * Z, net + ".0.weight", dx, lddx, B);
        g.accumulate = acc;
        return gemm(dt, g, s);
    }

    int forward(const float* images, const int64_t* ctx, const int64_t* post, const int64_t* tgt, const float* eps, int p2,
                uint64_t seed_, hipStream_t s) {
        BLT_REQUIRE(bound, "engine_forward: engine not bound");
        BLT_REQUIRE((images || !regions) && ctx && post && tgt, "engine_forward: null input");
        BLT_REQUIRE(!p2 || eps, "engine_forward: eps required in phase 2");
        phase2 = p2; seed = seed_; fwd_done = false;
        stamp(0, s);
        // with an optimiser update still in flight only the streams that read trainable parameters wait for it (below)
        const bool overlap_opt = opt_is_pending() && use_streams;
        if (opt_is_pending() && !overlap_opt) { RC(sync_opt(s)); opt_mark_synced(); }
        // loss statistics [0..3]; [4] (gradient norm) belongs to the optimiser, which may still be reading it
        if (hipMemsetAsync(stats - 1, 0, 5 * sizeof(float), s) != hipSuccess) { blt_set_error("engine_forward: memset failed"); return BLT_ERR_HIP; }
        if (overlap_opt) {
            // The frozen CNN does not depend on the update: it is enqueued first (unless it ran ahead: then the encoders' launches are),
            // everything else goes behind the optimiser — the token / encoder streams behind its FIRST stage only (embedding + encoder
            // parameters), the caller's stream behind the whole update, where the decoder-side weight shadows and the gradient memset
            // then run underneath the encoder stacks.
            hipStream_t s0 = side[0];
            const bool staged = steps->opt_stage1_ev != nullptr && blt_debug_get(20) != 1;
            const bool derive = shadows_current();
            RC(fork(s, s0, fj[0]));
            const bool ahead = pf_n > 0;
            if (!ahead) { RC(cnn_fwd(images, s)); stamp(11, s); }
            if (staged) {
                if (hipStreamWaitEvent(s0, steps->opt_stage1_ev, 0) != hipSuccess) { blt_set_error("engine_forward: optimiser wait failed"); return BLT_ERR_HIP; }
                RC(shadows(s0, 1, derive));
            } else {
                RC(sync_opt(s0));
                RC(shadows(s0, 3, derive));
                if (bn_train) RC(zero_grads_early(s0));
            }
            RC(forward_tokens(ctx, post, tgt, s0, s0, false));
            // Stage-2 consumers (decoder-side shadows, gradient memset) need the WHOLE update.  side[1] is the optimiser's own stream, so
            // work enqueued there is behind it by stream order: with the conv stack inline on `s` (the long pole of this phase) they go
            // there, in front of the context encoder (the shorter of the two stacks); with the stack run ahead `s` has nothing else to do.
            if (staged && !ahead) {
                RC(sync_opt(side[1]));      // (an engine that shares the parameters with the one whose stream ran the update is not behind it)
                RC(shadows(side[1], 2, derive));
                if (bn_train) RC(zero_grads_early(side[1]));
            }
            RC(fork(s0, side[1], fj[1]));
            RC(stack_fwd(enc, nullptr, nullptr, side[1]));
            if (run_renc()) RC(stack_fwd(renc, nullptr, nullptr, s0));
            if (ahead) { RC(cnn_fwd(images, s)); stamp(11, s); }      // (waits for the prefetched feature and the whole update, then the head)
            if (staged && ahead) {
                RC(shadows(s, 2, derive));
                if (bn_train) RC(zero_grads_early(s));
            }
            shadows_done(derive);
            RC(fork(s0, s, fj[2]));
            RC(fork(side[1], s, fj[3]));
            opt_mark_synced();            // s is now ordered behind the update
            return forward_tail(eps, s);
        }
        // weight shadows
        hipStream_t s0 = use_streams ? side[0] : s, s1 = use_streams ? side[1] : s;
        RC(forward_tokens(ctx, post, tgt, s, s0));
        if (use_streams && bn_train) RC(zero_grads_early(s0));      // s0 was forked from s (behind any earlier optimiser update) in forward_tokens
        // three independent sub-graphs until the image feature is injected: CNN (main) | embedding + posterior encoder (side 0) |
        // context encoder (side 1)
        if (use_streams) RC(fork(s0, s1, fj[1]));
        // the CNN is the long pole (1.3 ms of the step) and its ~90 launches take the host ~0.3 ms to enqueue: it goes first when the
        // side streams exist, so that the GPU is not left waiting for it behind the (short) encoder stacks' enqueue
        if (use_streams) { RC(cnn_fwd(images, s)); stamp(11, s); }      // [11] image feature done (CNN stream, before the encoders are joined)
        RC(stack_fwd(enc, nullptr, nullptr, s1));
        if (run_renc()) RC(stack_fwd(renc, nullptr, nullptr, s0));
        if (!use_streams) RC(cnn_fwd(images, s));
        if (use_streams) { RC(fork(s0, s, fj[2])); RC(fork(s1, s, fj[3])); }
        return forward_tail(eps, s);
    }

    // weight shadows, token preparation (on `s`), then the shared embedding of the three token streams (on `se`, forked from `s`
    // unless they are the same stream)
    // bf16 shadows of the GEMM weights (plain + transposed; biases / LayerNorm / embedding rows are read in fp32).  parts: bit 0 = the
    // parameters the token / encoder path reads (two encoder stacks, shared embedding), bit 1 = the rest (decoder, vocabulary projection,
    // reconstructor, latent-phase heads); 3 = everything, one launch.  derive: only the transposed copies, from the plain shadow the
    // optimiser pass keeps current (shadows_current()); else both are rebuilt from the fp32 parameters.
    bool shadows_current() const { return trust_shadows && shadow_gen == steps->params_gen; }
    void shadows_done(bool derive) { if (!derive) shadow_gen = steps->params_gen; }      // the plain shadow mirrors the parameters as of now
    int shadow_range(int e0, int e1, bool derive, hipStream_t s) {
        if (e1 <= e0) return BLT_OK;
        const int t0 = tlist[e0].tile0, t1 = (e1 < (int)tlist.size()) ? tlist[e1].tile0 : ttiles;
        if (t1 <= t0) return BLT_OK;
        const char* tab = (const char*)ttable + (size_t)e0 * sizeof(TEnt);
        if (derive) return blt_shadow_transpose_bf16(wshadow, wshadowT, tab, e1 - e0, t1 - t0, s, t0);
        return blt_shadow_transpose(train, wshadow, wshadowT, tab, e1 - e0, t1 - t0, s, t0);
    }
    int shadows(hipStream_t s, int parts, bool derive) {
        if (dt != BLT_BF16 || tlist.empty()) return BLT_OK;
        const int n = (int)tlist.size();
        // (the gamma-folded operands W' / s / c of the LayerNorm consumers are rebuilt from the fp32 parameters with the same parts)
        RC(fold_prepare(parts, s));
        if (parts == 3) return shadow_range(0, n, derive, s);
        if (parts & 1) { RC(shadow_range(t_dec_n, t_main_n, derive, s)); RC(shadow_range(t_heads_n, n, derive, s)); }
        if (parts & 2) { RC(shadow_range(0, t_dec_n, derive, s)); RC(shadow_range(t_main_n, t_heads_n, derive, s)); }
        return BLT_OK;
    }

    // The posterior encoder's output feeds the latent layer only (encoder_transformer.py:33-35): while `latent_transformer` is off
    // (train_iq.py:108-111, the first num_pretraining_steps steps) the reference still runs r_encoder and drops the result
    // (encoder_transformer.py:23-25) — no output, loss, gradient or statistic depends on it, and dropout here is addressed by site id, so
    // skipping it shifts no random stream.  debug key 23 bit 0: run it anyway (A/B and the bit-identity test).
    bool run_renc() const { return phase2 || (blt_debug_get(23) & 1); }

    int forward_tokens(const int64_t* ctx, const int64_t* post, const int64_t* tgt, hipStream_t s, hipStream_t se, bool with_shadows = true) {
        if (with_shadows) {
            const bool derive = shadows_current();
            RC(shadows(s, 3, derive));
            shadows_done(derive);
        }
        if (wemb_pad) RC(blt_cast_rows(BLT_F32, P("embedding.1.weight"), E, dt, wemb_pad, ld_wemb, H, E, s));
        RC(blt_prep_tokens((const long long*)ctx, (const long long*)post, (const long long*)tgt, B, Sa, Sp, T, ids_all, pos_all, tgt_shift,
                           tgt32, ctx32, post32, counters, V, stats - 1, s));
        if (se != s) RC(fork(s, se, fj[0]));
        // shared embedding over the three token streams at once (iq.py:72-78): gather -> Linear(E,H) + bias + timing signal
        // (token rows are ordered context | target | posterior: without the posterior encoder its rows are not embedded either)
        const int Memb = run_renc() ? Mtot : Ma + Mt;
        RC(blt_embed_gather(dt, P("embedding.0.weight"), ids_all, emb_rows, Memb, E, Epad, se));
        {
            int ldw;
            const void* w = W("embedding.1.weight", &ldw);
            GemmArgs g = mk(emb_rows, Epad, 0, w, ldw, 0, X_all, H, Memb, H, E);
            g.bias = P("embedding.1.bias");
            g.rowtab = timing; g.rowidx = pos_all; g.ldt = H;
            if (fold_on()) {      // the first LayerNorm of every stack is folded into its q|k|v projection
                set_stat(g, stat_emb);
                enc.stat_in_parts = renc.stat_in_parts = dec.stat_in_parts = stat_parts();
            }
            RC(gemm(dt, g, se));
        }
        enc.x_in = X_all; enc.key_ids = ctx32;
        dec.x_in = (char*)X_all + (size_t)Ma * H * es; dec.key_ids = tgt_shift;
        renc.x_in = (char*)X_all + (size_t)(Ma + Mt) * H * es; renc.key_ids = post32;
        return BLT_OK;
    }

    // target_embedding[:,0] += image_features (+ z) (decoder_transformer.py:31,34); with folded LayerNorms the row sums of those B rows
    // (left by the embedding GEMM) are replaced by those of the new rows
    int dec_row0_add(const void* z, hipStream_t s) {
        if (fold_on()) return blt_rows_add_stat(dt, dec.x_in, (long)T * H, feats, H, z, H, B, H, 1, dec.stat_in, 2L * stat_slots * T, dec.stat_in_parts, s);
        return blt_rows_add(dt, dec.x_in, (long)T * H, feats, H, z, H, B, H, 1, s);
    }

    // everything after the image feature exists: latent, decoder, vocabulary projection, reconstructor (all on `s`)
    int forward_tail(const float* eps, hipStream_t s) {
        stamp(1, s);           // CNN + both encoders done (the side streams were joined just before)
        kv_hoisted = false;
        RC(blt_rows_add(dt, enc.out, (long)Sa * H, feats, H, nullptr, 0, B, H, 1, s));   // encoder_outputs[:,0] += image_features
        // Branch stream: everything below that the decoder stack does not need (z_classifier, image reconstructor) or needs only
        // later (the encoder-side key/value projections of its layers) leaves the critical path; joined at the end of forward.
        hipStream_t sb = use_streams ? side[0] : s;
        if (phase2) {
            if (hipMemcpyAsync(eps_dev, eps, sizeof(float) * (size_t)B * Z, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                blt_set_error("engine_forward: eps copy failed");
                return BLT_ERR_HIP;
            }
            // Latent.forward (transformer_layers.py:41-59): prior(x), posterior(cat(x_p, x))
            RC(blt_copy2d(dt, enc.out, Sa * H, (char*)cat_in + (size_t)H * es, 2 * H, B, H, s));
            RC(blt_copy2d(dt, renc.out, Sp * H, cat_in, 2 * H, B, H, s));
            if (sb != s) { RC(fork(s, sb, fj[11])); RC(dec_kv_fwd(sb)); }
            RC(mlp3_fwd("latent_layer.mean_logvar_prior", (char*)cat_in + (size_t)H * es, 2 * H, mlvp_h1, mlvp_h2, mlvp, s));
            RC(mlp3_fwd("latent_layer.mean_logvar_posterior", cat_in, 2 * H, mlvq_h1, mlvq_h2, mlvq, s));
            RC(blt_latent_fwd(dt, mlvp, mlvq, eps_dev, zlat, stats + 2, B, Z, 2 * Z, s));
            RC(gemm(dt, lin(zlat, Z, "latent_projection.weight", "latent_projection.bias", zproj, H, B), s));
            // target_embedding[:,0] += image_features + z ; z_logit = z_classifier(z + image_features)
            RC(dec_row0_add(zproj, s));
            if (sb != s) RC(fork(s, sb, fj[12]));            // the branch continues behind z
            RC(blt_rows_add(dt, zc_in, H, feats, H, zproj, H, B, H, 0, sb));
            RC(gemm(dt, lin(zc_in, H, "decoder.z_classifier.weight", "decoder.z_classifier.bias", zlogit, ldV, B), sb));
            RC(blt_rows_add(dt, r_in, H, enc.out, (long)Sa * H, zproj, H, B, H, 0, sb));
        } else {
            RC(dec_row0_add(nullptr, s));
            if (sb != s) { RC(fork(s, sb, fj[11])); RC(dec_kv_fwd(sb)); }
            RC(blt_copy2d(dt, enc.out, Sa * H, r_in, H, B, H, sb));
        }
        // image_reconstructor (mlp.py:49-56): Linear, ReLU, Dropout(0), Linear
        {
            GemmArgs g = lin(r_in, H, "image_reconstructor.layers.fc0.weight", "image_reconstructor.layers.fc0.bias", hrec, F, B);
            g.relu = 1;
            RC(gemm(dt, g, sb));
            RC(gemm(dt, lin(hrec, F, "image_reconstructor.layers.fc1.weight", "image_reconstructor.layers.fc1.bias", recon, H, B), sb));
        }
        stamp(2, s);           // latent / injections done, decoder starts
        RC(stack_fwd(dec, enc.out, ctx32, s));
        stamp(3, s);
        kv_hoisted = false;
        RC(gemm(dt, lin(dec.out, H, "decoder.output.weight", "decoder.output.bias", logits, ldV, Mt), s));
        if (sb != s) RC(fork(sb, s, fj[14]));
        stamp(4, s);           // vocabulary projection done: end of forward
        fwd_done = true;
        return BLT_OK;
    }

    // ---------------------------------------------------------------------------------------------
    // IQ.decode_greedy (reference models/iq.py:117-152): encoder once, then T decoder passes over the growing prefix.  Each pass
    // runs the decoder on all T positions with "future keys do not exist" attention (causal = 2), which gives position t exactly
    // the value the reference computes on the length-(t+1) prefix; the next token is the argmax of that position's logits.
    // ---------------------------------------------------------------------------------------------
    int decode_greedy(const float* images, const int64_t* ctx, const float* eps, int p2, int train_bn, int* tokens, int* top_idx, float* top_val,
                      hipStream_t s) {
        BLT_REQUIRE(bound, "engine_decode_greedy: engine not bound");
        BLT_REQUIRE((images || !regions) && ctx && tokens && top_idx && top_val, "engine_decode_greedy: null pointer");
        BLT_REQUIRE(!p2 || eps, "engine_decode_greedy: eps required when the latent path is on");
        BLT_REQUIRE(c.attention_dropout == 0.f && c.relu_dropout == 0.f, "engine_decode_greedy: create the decode engine with dropout 0");
        if (opt_is_pending()) { RC(sync_opt(s)); opt_mark_synced(); }
        phase2 = p2; seed = 0; fwd_done = false;
        const bool saved_bn = bn_train, saved_streams = use_streams;
        bn_train = train_bn != 0; use_streams = false; causal_mode = 2;
        int rc = decode_body(images, ctx, eps, tokens, top_idx, top_val, s);
        bn_train = saved_bn; use_streams = saved_streams; causal_mode = 1;
        return rc;
    }
    int decode_body(const float* images, const int64_t* ctx, const float* eps, int* tokens, int* top_idx, float* top_val, hipStream_t s) {
        if (dt == BLT_BF16) RC(blt_cast_rows(BLT_F32, train, (int)1, BLT_BF16, wshadow, 1, tsize, 1, s));
        RC(fold_prepare(3, s));
        if (wemb_pad) RC(blt_cast_rows(BLT_F32, P("embedding.1.weight"), E, dt, wemb_pad, ld_wemb, H, E, s));
        if (hipMemsetAsync(stats - 1, 0, sizeof(float), s) != hipSuccess) { blt_set_error("engine_decode_greedy: memset failed"); return BLT_ERR_HIP; }
        RC(blt_prep_decode((const long long*)ctx, B, Sa, T, ids_all, pos_all, ctx32, V, stats - 1, s));
        int* ys = ids_all + Ma;
        RC(cnn_fwd(images, s));
        int ldw;
        const void* we = W("embedding.1.weight", &ldw);
        auto embed = [&](int row0, int rows) -> int {
            RC(blt_embed_gather(dt, P("embedding.0.weight"), ids_all + row0, (char*)emb_rows + (size_t)row0 * Epad * es, rows, E, Epad, s));
            GemmArgs g = mk((char*)emb_rows + (size_t)row0 * Epad * es, Epad, 0, we, ldw, 0, (char*)X_all + (size_t)row0 * H * es, H, rows, H, E);
            g.bias = P("embedding.1.bias");
            g.rowtab = timing; g.rowidx = pos_all + row0; g.ldt = H;
            if (fold_on()) {
                set_stat(g, stat_emb + 2 * (size_t)stat_slots * row0);
                (row0 == 0 ? enc : dec).stat_in_parts = stat_parts();
            }
            return gemm(dt, g, s);
        };
        RC(embed(0, Ma));
        enc.x_in = X_all; enc.key_ids = ctx32;
        dec.x_in = (char*)X_all + (size_t)Ma * H * es; dec.key_ids = ys;
        RC(stack_fwd(enc, nullptr, nullptr, s));
        RC(blt_rows_add(dt, enc.out, (long)Sa * H, feats, H, nullptr, 0, B, H, 1, s));
        if (phase2) {
            // Latent.forward with x_p = None (transformer_layers.py:41-48): z = eps * exp(0.5 logvar_prior) + mean_prior
            if (hipMemcpyAsync(eps_dev, eps, sizeof(float) * (size_t)B * Z, hipMemcpyDeviceToDevice, s) != hipSuccess) return BLT_ERR_HIP;
            RC(blt_copy2d(dt, enc.out, Sa * H, (char*)cat_in + (size_t)H * es, 2 * H, B, H, s));
            RC(mlp3_fwd("latent_layer.mean_logvar_prior", (char*)cat_in + (size_t)H * es, 2 * H, mlvp_h1, mlvp_h2, mlvp, s));
            RC(blt_latent_fwd(dt, mlvp, mlvp, eps_dev, zlat, stats + 6, B, Z, 2 * Z, s));      // reparameterise with the PRIOR; KL slot unused
            RC(gemm(dt, lin(zlat, Z, "latent_projection.weight", "latent_projection.bias", zproj, H, B), s));
        }
        // the decoder layers' encoder-side key / value projections read encoder_outputs only (transformer_layers.py:330-342: the reference
        // recomputes them in every one of its prefix re-decodes, iq.py:134-141): once per batch, not once per step (debug key 28 = 1: per step)
        // debug key 29 = 1: the round-2 form, one decoder pass over all T rows per step (A/B and the equivalence test); it may also
        // recompute the hoisted projections per step (key 28 = 1)
        const bool incremental = blt_debug_get(29) != 1;
        const bool hoist = incremental || blt_debug_get(28) != 1;
        if (hoist) RC(dec_kv_fwd(s));
        for (int t = 0; t < T; ++t) {
            RC(embed(Ma, Mt));
            RC(dec_row0_add(phase2 ? zproj : nullptr, s));     // [:,0] += z + image_features
            const int rc = incremental ? dec_step_fwd(t, ctx32, s) : stack_fwd(dec, enc.out, ctx32, s);
            if (rc) { kv_hoisted = false; return rc; }
            GemmArgs g = lin((char*)dec.out + (size_t)t * H * es, T * H, "decoder.output.weight", "decoder.output.bias", zlogit, ldV, B);
            RC(gemm(dt, g, s));
            RC(blt_argmax_top6(dt, zlogit, ldV, B, V, t, T, ys, tokens, top_idx, top_val, s));
        }
        kv_hoisted = false;
        return BLT_OK;
    }

    // ---------------------------------------------------------------------------------------------
    // backward pieces.  `dx` holds the gradient w.r.t. the layer OUTPUT on entry and w.r.t. its INPUT on exit.
    // ---------------------------------------------------------------------------------------------
    // dx_in = d(sub-layer output); dx_out = d(sub-layer input) = LayerNorm backward of the FFN branch + dx_in (residual); over the rows of `rw`
    int ffn_bwd(const std::string& fp_, const void* xn, const void* xres, const float* m, const float* r, const std::string& ln,
                Layer& y, const void* dx_in, void* dx_out, const Rows& rw, int k, hipStream_t s) {
        void* gB = sB[k];
        const float ks = relu_ks();
        // y.gY = d(FFN output) through the ReLU + dropout: already written by the LayerNorm backward that produced dx_in
        RC(wgrad_later(y.gY, rw.ldH, y.h, rw.ldF, fp_ + "layers.1.weight", (fp_ + "layers.1.bias").c_str(), rw.M, s));
        GemmArgs g = dgrad(y.gY, rw.ldH, fp_ + "layers.1.weight", y.gF, rw.ldF, rw.M);
        g.maskY = y.h; g.ldm = rw.ldF; g.mask_scale = ks;
        RC(gemm(dt, g, s));
        RC(gemm(dt, dgrad(y.gF, rw.ldF, fp_ + "layers.0.weight", gB, rw.ldH, rw.M), s));
        // (folded LayerNorm: its output xn — the X operand of the first Linear's weight gradient — is written by THIS backward launch)
        RC(ln_bwd(gB, xres, ln, m, r, dx_in, dx_out, rw.M, s, nullptr, 1.f, nullptr, rw.ldH, fold_on() ? const_cast<void*>(xn) : nullptr));
        return wgrad_later(y.gF, rw.ldF, xn, rw.ldH, fp_ + "layers.0.weight", (fp_ + "layers.0.bias").c_str(), rw.M, s);
    }

    // `dx` holds d(stack output before the final LayerNorm) on entry and d(stack input) on exit; the gradients in between live in the
    // layers' own buffers (Layer::dx1..3)
    float relu_ks() const { return (c.relu_dropout > 0.f) ? 1.f / (1.f - c.relu_dropout) : 1.f; }
    int stack_bwd(Stack& st, void* dx, const void* enc_out, const int* src_ids, hipStream_t s, const std::vector<int>* flush_plan = nullptr) {
        const int M = st.M, S = st.S;
        void *gA = sA[st.scr], *gB = sB[st.scr], *gC = sC[st.scr];
        const void* cur = dx;
        for (int l = L - 1; l >= 0; --l) {
            Layer& y = st.layers[l];
            const void* x = (l == 0) ? st.x_in : st.layers[l - 1].x2;
            const std::string lp = st.prefix + (st.dec ? ".dec." : ".enc.") + std::to_string(l) + ".";
            const std::string a1 = lp + (st.dec ? "multi_head_attention_dec." : "multi_head_attention.");
            const std::string ln1 = lp + (st.dec ? "layer_norm_mha_dec" : "layer_norm_mha");
            const Rows rw = rows_of(st, l);
            // row-0-only top layer: its row-wise backward touches B strided rows of d(sub-layer output) / d(attention context); the attention
            // core below and the residual term of the first LayerNorm's backward read ALL rows, zero everywhere else (Stack::gA_top)
            void* dx1 = rw.sub ? st.dx1_top : y.dx1;
            void* gAo = rw.sub ? st.gA_top : gA;      // d(self-attention context)
            if (st.dec) {
                RC(ffn_bwd(lp + "positionwise_feed_forward.", y.xn3, y.x1b, y.m3, y.r3, lp + "layer_norm_ffn", y, cur, y.dx1, rw, st.scr, s));
                cur = y.dx1;
                // encoder-decoder attention
                const std::string a2 = lp + "multi_head_attention_enc_dec.";
                RC(wgrad_later(cur, H, y.ctx2, H, a2 + "output_linear.weight", nullptr, M, s));
                RC(gemm(dt, dgrad(cur, H, a2 + "output_linear.weight", gA, H, M), s));
                RC(attn_bwd(y.q2, H, y.kv2, (char*)y.kv2 + (size_t)H * es, 2 * H, gA, y.gQ, H, y.gKV, (char*)y.gKV + (size_t)H * es, 2 * H, src_ids,
                            S, Sa, 0, sid(st.id, l, 3), s));
                RC(gemm(dt, dgrad(y.gQ, H, a2 + "query_linear.weight", gC, H, M), s));
                {   // key/value projections of encoder_outputs: [2H,H] fused
                    const PInfo& pk = tpi(a2 + "key_linear.weight");
                    GemmArgs g = mk(y.gKV, 2 * H, 1, enc_out, H, 1, grad + pk.off, H, 2 * H, H, Ma);
                    g.out_f32 = 1; g.split_k = 32;
                    RC(wgrad_later(g, s));
                    g = dgrad_rows(y.gKV, 2 * H, a2 + "key_linear.weight", 2 * H, d_enc, H, Ma);
                    g.accumulate = (l == L - 1) ? 0 : 1;
                    RC(gemm(dt, g, s));
                }
                const std::string ln2 = lp + "layer_norm_mha_enc";
                RC(ln_bwd(gC, y.x1, ln2, y.m2, y.r2, cur, y.dx2, M, s, nullptr, 1.f, nullptr, 0, fold_on() ? y.xn2 : nullptr));
                RC(wgrad_later(y.gQ, H, y.xn2, H, a2 + "query_linear.weight", nullptr, M, s));
                cur = y.dx2;
            } else {
                RC(ffn_bwd(lp + "positionwise_feed_forward.", y.xn2, y.x1, y.m2, y.r2, lp + "layer_norm_ffn", y, cur, dx1, rw, st.scr, s));
                cur = dx1;
            }
            // self attention: output projection over the rows of `rw`, the attention core and the q|k|v projection over all rows
            RC(wgrad_later(cur, rw.ldH, y.ctx, rw.ldH, a1 + "output_linear.weight", nullptr, rw.M, s));
            RC(gemm(dt, dgrad(cur, rw.ldH, a1 + "output_linear.weight", gAo, rw.ldH, rw.M), s));
            RC(attn_bwd(y.qkv, 3 * H, (char*)y.qkv + (size_t)H * es, (char*)y.qkv + (size_t)2 * H * es, 3 * H, gAo, y.gQKV, 3 * H,
                        (char*)y.gQKV + (size_t)H * es, (char*)y.gQKV + (size_t)2 * H * es, 3 * H, st.key_ids, S, S, st.dec ? 1 : 0, sid(st.id, l, 0), s));
            RC(gemm(dt, dgrad_rows(y.gQKV, 3 * H, a1 + "query_linear.weight", 3 * H, gB, H, M), s));
            // the last sub-layer of the stack writes d(stack input) back into the caller's buffer (its old content is dead by now:
            // only the top layer's FFN branch read it, on this same stream)
            void* out = (l == 0) ? dx : (st.dec ? y.dx3 : y.dx2);
            // ... and, for the layer below, the masked gradient that opens its FFN backward (ffn_bwd)
            const void* nmask = (l > 0) ? st.layers[l - 1].y2 : nullptr;
            void* ngY = (l > 0) ? st.layers[l - 1].gY : nullptr;
            RC(ln_bwd(gB, x, ln1, y.m1, y.r1, cur, out, M, s, nmask, relu_ks(), ngY, 0, fold_on() ? y.xn1 : nullptr));
            {   // q|k|v weight gradient: its X operand is this LayerNorm's output
                const PInfo& pq = tpi(a1 + "query_linear.weight");
                GemmArgs g = mk(y.gQKV, 3 * H, 1, y.xn1, H, 1, grad + pq.off, H, 3 * H, H, M);
                g.out_f32 = 1; g.split_k = 32;
                RC(wgrad_later(g, s));
            }
            cur = out;
            // in-stack flush points (build_params: groups of whole layers of >= ~32 MB of parameters): the collected weight gradients go
            // to the weight-gradient stream now, run under the rest of THIS chain and close a gradient bucket, whose all-reduce (N > 1)
            // can start while backward is still running.  only while group_flush is on (a data-parallel exchange exists)
            // (debug key 21, A/B: bit mask of the stacks that flush in-stack although no exchange asked for it: 1 decoder, 2 context, 4 posterior)
            const bool forced = (blt_debug_get(21) >> (st.dec ? 0 : (st.id == 0 ? 1 : 2))) & 1;
            if (defer_wgrads && ((group_flush && blt_debug_get(19) != 1) || forced) && flush_plan && (*flush_plan)[l] >= 0) {      // (key 19 = 1: A/B, no in-stack flush)
                // The decoder's first flush carries the reconstructor's / z_classifier's weight gradients too, whose operands (d_recon, g_rec1,
                // dzl) are written on the BRANCH stream side[0] (backward_core); `s` joins that branch only behind the decoder chain, so the
                // weight-gradient stream orders itself behind the branch here (fj[12] was recorded on it before this chain started)
                if (st.dec && use_streams && hipStreamWaitEvent(side[1], fj[12], 0) != hipSuccess) { blt_set_error("backward: stream join failed"); return BLT_ERR_HIP; }
                RC(flush_wgrads(s, side[1], fj[6], (*flush_plan)[l]));
            }
        }
        return BLT_OK;
    }

    // CNN head: BatchNorm1d -> fc backward in fp32 (the backbone itself is frozen, encoder_cnn.py:18-19); bias gradient from the GEMM's dY
    // staging registers
    int cnn_head_bwd(hipStream_t s) {
        RC(blt_cast_rows(dt, d_feats, H, BLT_F32, dfeats32, H, B, H, s));
        RC(blt_bn1d_bwd(BLT_F32, dfeats32, featpre, P("encoder_cnn.bn.weight"), bn1_mean, bn1_rstd, dfeatpre32, G("encoder_cnn.bn.weight"),
                        G("encoder_cnn.bn.bias"), B, H, s));
        if (region_attn) {
            const int BR = B * c.num_regions;
            RC(blt_region_attn_bwd(dt, Pr, P("encoder_cnn.region_attn.weight"), ralpha, dfeatpre32, dPr, G("encoder_cnn.region_attn.weight"), B,
                                   c.num_regions, H, s));
            return wgrad(dPr, H, xr, FD, fcw, fcb.c_str(), BR, s);      // dW = dP^T X, db = column sums of dP (the regions have no gradient)
        }
        const PInfo& pw = tpi(fcw);
        GemmArgs g = mk(dfeatpre32, H, 1, pooled, FD, 1, grad + pw.off, FD, H, FD, B);
        g.accumulate = 1; g.a_rowsum = G(fcb);
        return gemm(BLT_F32, g, s);
    }

    // Everything downstream of the loss-gradient seeds: `logits` holds d(output), dzl holds d(z_logit) (phase 2),
    // d_feats holds the direct gradient of image_features, d_recon that of the reconstruction.
    int backward_core(float kld_g, hipStream_t s) {
        pending_wgrads.clear();
        pending_ln.clear();
        ln_pool_used = 0;
        flush_idx = 0;
        defer_wgrads = false;
        // every weight gradient from here on is collected and issued on a side stream at the next flush point (their operands live
        // in buffers that nothing overwrites during this backward pass); only the input-gradient chain stays on `s`
        defer_wgrads = use_streams;
        // ---- branch stream: the reconstructor's and z_classifier's input gradients depend on the loss kernels only; they run on side[0]
        // under the decoder chain and are summed into the row-0 gradients behind it (row0_sums) ----
        hipStream_t sbr = use_streams ? side[0] : s;
        if (sbr != s) RC(fork(s, sbr, fj[15]));
        RC(wgrad_later(d_recon, H, hrec, F, "image_reconstructor.layers.fc1.weight", "image_reconstructor.layers.fc1.bias", B, s));
        {
            GemmArgs g = dgrad(d_recon, H, "image_reconstructor.layers.fc1.weight", g_rec1, F, B);
            g.maskY = hrec; g.ldm = F; g.mask_scale = 1.f;
            RC(gemm(dt, g, sbr));
        }
        RC(wgrad_later(g_rec1, F, r_in, H, "image_reconstructor.layers.fc0.weight", "image_reconstructor.layers.fc0.bias", B, s));
        RC(gemm(dt, dgrad(g_rec1, F, "image_reconstructor.layers.fc0.weight", g_rin, H, B), sbr));   // d r_in
        if (phase2) {
            RC(wgrad_later(dzl, ldV, zc_in, H, "decoder.z_classifier.weight", "decoder.z_classifier.bias", B, s));
            RC(dgrad_bigk(dzl, ldV, "decoder.z_classifier.weight", g_zc, H, B, sbr, acc_big2, (size_t)B * H));
        }
        if (sbr != s && hipEventRecord(fj[12], sbr) != hipSuccess) { blt_set_error("backward: event record failed"); return BLT_ERR_HIP; }
        // ---- vocabulary projection + decoder ----
        RC(wgrad_later(logits, ldV, dec.out, H, "decoder.output.weight", "decoder.output.bias", Mt, s));
        void* gA = sA[0];
        RC(dgrad_bigk(logits, ldV, "decoder.output.weight", gA, H, Mt, s));
        void* dxT = (char*)dX_all + (size_t)Ma * H * es;
        {
            const void* xL = dec.layers[L - 1].x2;
            RC(ln_bwd(gA, xL, "decoder.decoder.layer_norm", dec.mF, dec.rF, nullptr, dxT, Mt, s, dec.layers[L - 1].y2, relu_ks(),
                      dec.layers[L - 1].gY));
        }
        stamp(6, s);           // vocabulary input gradient + final LayerNorm backward done
        RC(stack_bwd(dec, dxT, enc.out, ctx32, s, &dec_flush));
        stamp(7, s);
        if (sbr != s && hipStreamWaitEvent(s, fj[12], 0) != hipSuccess) { blt_set_error("backward: stream join failed"); return BLT_ERR_HIP; }
        // target_embedding[:,0] += image_features (+ z); r_in = encoder row 0 (+ z); zc_in = z + image_features
        RC(blt_row0_sums(dt, dxT, (long)T * H, g_rin, phase2 ? g_zc : nullptr, d_feats, phase2 ? d_zproj : nullptr, d_enc, (long)Sa * H, B, H, s));
        if (phase2) {
            // ---- latent projection, reparameterisation + KL, prior / posterior nets ----
            RC(wgrad_later(d_zproj, H, zlat, Z, "latent_projection.weight", "latent_projection.bias", B, s));
            RC(gemm(dt, dgrad(d_zproj, H, "latent_projection.weight", g_b3, Z, B), s));   // dz
            RC(blt_latent_bwd(dt, mlvp, mlvq, eps_dev, g_b3, kld_g, g_b4, g_mq, B, Z, 2 * Z, s));
            // posterior net: d cat(x_p, x)
            RC(mlp3_bwd("latent_layer.mean_logvar_posterior", 0, cat_in, 2 * H, mlvq_h1, mlvq_h2, g_mq, g_cat, 2 * H, 0, s));
            // prior net: d x accumulated into the x half of d cat
            RC(mlp3_bwd("latent_layer.mean_logvar_prior", 1, (char*)cat_in + (size_t)H * es, 2 * H, mlvp_h1, mlvp_h2, g_b4,
                        (char*)g_cat + (size_t)H * es, 2 * H, 1, s));
            RC(blt_rows_add(dt, d_enc, (long)Sa * H, (char*)g_cat + (size_t)H * es, 2 * H, nullptr, 0, B, H, 1, s));
        }
        // encoder_outputs[:,0] += image_features
        RC(blt_rows_add(dt, d_feats, H, d_enc, (long)Sa * H, nullptr, 0, B, H, 1, s));
        // The decoder's weight gradients (one grouped launch: six rounds of the whole chip) go to side[1] only HERE, behind the latent
        // section: its dozen [B, *]-sized launches sit on the critical path between the decoder chain and the two encoder chains, and
        // launched beside the grouped kernel they waited for its 40 us workgroups to free a CU one by one (0.55 ms instead of 0.1).  The
        // latent nets' own weight gradients ride in the same launch; bucket 0 (decoder.*) is complete when side[1] gets past it.
        RC(flush_wgrads(s, use_streams ? side[1] : s, fj[6], bk_dec_last, phase2 ? bk_late0 : -1));
        if (!group_flush || blt_debug_get(19) == 1) RC(record_buckets(dec_flush, use_streams ? side[1] : s));
        // d(image feature) is final here, long before the encoder chains are: the CNN head's backward (BatchNorm1d -> fc) goes to the
        // weight-gradient stream now instead of closing the chain
        if (use_streams) { RC(fork(s, side[1], fj[13])); RC(cnn_head_bwd(side[1])); }
        int Memb = Ma + Mt;
        hipStream_t s0 = (use_streams && phase2) ? side[0] : s;
        if (phase2) {
            // ---- posterior encoder (side stream, own scratch set): only row 0 of its output carries gradient ----
            if (s0 != s) RC(fork(s, s0, fj[4]));
            // (row-0-only top layer: the final LayerNorm's backward reads and writes those B rows only, nothing reads the others)
            const Rows rwP = rows_of(renc, L - 1);
            if (!rwP.sub && hipMemsetAsync(d_renc, 0, (size_t)Mp * H * es, s0) != hipSuccess) { blt_set_error("backward: memset failed"); return BLT_ERR_HIP; }
            RC(blt_rows_add(dt, d_renc, (long)Sp * H, g_cat, 2 * H, nullptr, 0, B, H, 0, s0));
            void* dxP = (char*)dX_all + (size_t)(Ma + Mt) * H * es;
            const void* xL = renc.layers[L - 1].x2;
            RC(ln_bwd(d_renc, xL, "answer_encoder.r_encoder.layer_norm", renc.mF, renc.rF, nullptr, dxP, rwP.M, s0, renc.layers[L - 1].y2, relu_ks(),
                      renc.layers[L - 1].gY, rwP.ldH));
            RC(stack_bwd(renc, dxP, nullptr, nullptr, s0, &renc_flush));
            // the main stream will join s0 at THIS point (it needs the chain's result for the embedding backward); the posterior
            // encoder's weight gradients then run on s0 behind it, beside the context encoder's on side[1], and are joined at the very end
            if (s0 != s && hipEventRecord(fj[5], s0) != hipSuccess) { blt_set_error("backward: event record failed"); return BLT_ERR_HIP; }
            RC(flush_wgrads(s0, s0 != s ? s0 : side[1], fj[7], bk_renc_last));
            if (!group_flush || blt_debug_get(19) == 1) RC(record_buckets(renc_flush, s0 != s ? s0 : side[1]));
            Memb = Mtot;
        }
        // ---- context encoder (main stream) ----
        {
            const void* xL = enc.layers[L - 1].x2;
            RC(ln_bwd(d_enc, xL, "answer_encoder.encoder.layer_norm", enc.mF, enc.rF, nullptr, dX_all, Ma, s, enc.layers[L - 1].y2, relu_ks(),
                      enc.layers[L - 1].gY));
        }
        stamp(8, s);           // latent backward done, context-encoder backward starts (posterior encoder runs beside it)
        RC(stack_bwd(enc, dX_all, nullptr, nullptr, s, &enc_flush));
        stamp(9, s);
        RC(flush_wgrads(s, use_streams ? side[1] : s, fj[8], bk_enc_last));
        if (!group_flush || blt_debug_get(19) == 1) RC(record_buckets(enc_flush, use_streams ? side[1] : s));
        defer_wgrads = false;
        if (s0 != s && hipStreamWaitEvent(s, fj[5], 0) != hipSuccess) { blt_set_error("backward: stream join failed"); return BLT_ERR_HIP; }
        // ---- shared embedding (rows of the streams that received gradient): weight + bias gradient to the side stream, the gradient of
        // the embedding table stays here (it closes the chain) ----
        {
            const PInfo& pw = tpi("embedding.1.weight");
            GemmArgs g = mk(dX_all, H, 1, emb_rows, Epad, 1, grad + pw.off, E, H, E, Memb);
            g.out_f32 = 1; g.split_k = 32; g.a_rowsum = G("embedding.1.bias");
            defer_wgrads = use_streams;
            RC(wgrad_later(g, s));
            RC(flush_wgrads(s, side[1], fj[8]));
            defer_wgrads = false;
            int ldw;
            const void* w = W("embedding.1.weight", &ldw);
            RC(gemm(dt, mk(dX_all, H, 0, w, ldw, 1, dE, Epad, Memb, E, H), s));
            RC(blt_embed_scatter(dt, dE, Epad, ids_all, G("embedding.0.weight"), Memb, E, 0, s));
        }
        if (!use_streams) RC(cnn_head_bwd(s));
        if (use_streams) {      // join the weight-gradient streams
            RC(fork(side[1], s, fj[9]));
            if (s0 != s) RC(fork(s0, s, fj[10]));
        }
        // the tail bucket (embedding, CNN head) is final here; without side streams nothing was flushed in between: every bucket is
        for (size_t k = 0; k < buckets.size(); ++k)
            if (buckets[k].ev && ((int)k == bk_tail || !use_streams)) (void)hipEventRecord(buckets[k].ev, s);
        last_bwd_phase2 = phase2;
        stamp(10, s);          // every stream joined: end of backward
        return BLT_OK;
    }
    int zero_grads(hipStream_t s) {
        if (grads_zeroed) {          // done by forward() on a side stream: every gradient writer of this backward is forked from `s` later
            grads_zeroed = false;
            if (hipStreamWaitEvent(s, grad_zero_ev, 0) != hipSuccess) { blt_set_error("backward: grad memset wait failed"); return BLT_ERR_HIP; }
            return BLT_OK;
        }
        if (hipMemsetAsync(grad, 0, sizeof(float) * (size_t)tsize, s) != hipSuccess) { blt_set_error("backward: grad memset failed"); return BLT_ERR_HIP; }
        return BLT_OK;
    }

    int loss_backward(float kl_weight, hipStream_t s) {
        BLT_REQUIRE(bound && fwd_done, "engine_loss_backward: forward has not run");
        if (opt_is_pending()) { RC(sync_opt(s)); opt_mark_synced(); }
        RC(zero_grads(s));
        // train_iq.py:81-103.  The token cross-entropy (41 MB of logits) leads the decoder chain; the image-reconstruction MSE and the
        // bag-of-words CE only feed the branch stream's work (backward_core) and run there, beside it.
        hipStream_t sbr = use_streams ? side[0] : s;
        if (sbr != s) RC(fork(s, sbr, fj[11]));
        RC(blt_mse_fwd_bwd(dt, feats, recon, (long)B * H, c.image_recon_lambda, stats + 1, d_feats, d_recon, sbr, (long)B * H_true));
        float kld_g = 0.f;
        if (phase2) {
            RC(blt_bow_ce_fwd_bwd(dt, zlogit, ldV, tgt32, B, T, V, counters, c.aux_ceiling, stats + 3, dzl, sbr));
            kld_g = c.kl_ceiling * kl_weight;
        }
        RC(blt_ce_fwd_bwd(dt, logits, ldV, tgt32, Mt, V, counters, 1.f, stats + 0, 1, s));
        stamp(5, s);           // loss kernels enqueued
        pool_in_use = false;   // (the head's weight gradient, the last reader of the pooled slot, is enqueued by backward_core and joined into s)
        return backward_core(kld_g, s);
    }

    int backward_external(const float* d_output, const float* d_zlogit, float d_kld, const float* d_feats_in, const float* d_recon_in,
                          hipStream_t s) {
        BLT_REQUIRE(bound && fwd_done, "engine_backward_external: forward has not run");
        if (opt_is_pending()) { RC(sync_opt(s)); opt_mark_synced(); }
        RC(zero_grads(s));
        if (d_output) RC(blt_cast_rows(BLT_F32, d_output, V, dt, logits, ldV, Mt, V, s));
        else if (hipMemsetAsync(logits, 0, (size_t)Mt * ldV * es, s) != hipSuccess) return BLT_ERR_HIP;
        if (d_feats_in) RC(blt_cast_rows(BLT_F32, d_feats_in, H, dt, d_feats, H, B, H, s));
        else if (hipMemsetAsync(d_feats, 0, (size_t)B * H * es, s) != hipSuccess) return BLT_ERR_HIP;
        if (d_recon_in) RC(blt_cast_rows(BLT_F32, d_recon_in, H, dt, d_recon, H, B, H, s));
        else if (hipMemsetAsync(d_recon, 0, (size_t)B * H * es, s) != hipSuccess) return BLT_ERR_HIP;
        if (phase2) {
            if (d_zlogit) RC(blt_cast_rows(BLT_F32, d_zlogit, V, dt, dzl, ldV, B, V, s));
            else if (hipMemsetAsync(dzl, 0, (size_t)B * ldV * es, s) != hipSuccess) return BLT_ERR_HIP;
        }
        pool_in_use = false;
        return backward_core(d_kld, s);
    }

    // async_: run on the engine's optimiser stream behind `s_in` (see opt_stream); the next forward orders itself behind the update
    int optimizer_step(float lr, float max_norm, float b1, float b2, float eps, hipStream_t s_in, bool async_ = false) {
        BLT_REQUIRE(bound, "engine_optimizer_step: engine not bound");
        if (opt_is_pending()) { RC(sync_opt(s_in)); opt_mark_synced(); }
        hipStream_t s = s_in;
        if (async_) {
            RC(fork(s_in, opt_stream, opt_fork));
            s = opt_stream;
        }
        const int rc_ = optimizer_kernels(lr, max_norm, b1, b2, eps, s, async_);
        if (rc_) return rc_;
        if (async_) {
            if (hipEventRecord(opt_done, opt_stream) != hipSuccess) { blt_set_error("engine_optimizer_step: event record failed"); return BLT_ERR_HIP; }
            ++steps->opt_gen;                 // every engine sharing these buffers (this one included) now has an update to wait for
            steps->opt_done_ev = opt_done;
            steps->opt_stage1_ev = opt_stage1;
        } else {
            steps->opt_stage1_ev = nullptr;
        }
        return BLT_OK;
    }
    int optimizer_kernels(float lr, float max_norm, float b1, float b2, float eps, hipStream_t s, bool staged) {
        if (hipMemsetAsync(stats + 4, 0, sizeof(float), s) != hipSuccess) return BLT_ERR_HIP;
        // the latent-phase parameters sit right behind the others in the flat buffers: one launch covers both regions when both are live
        RC(blt_sumsq(grad, last_bwd_phase2 ? tsize : late_off, stats + 4, s));
        const int step_main = ++steps->main;
        if (last_bwd_phase2) ++steps->late;
        const int step_late = steps->late;
        // bf16 mode: the update also writes the plain weight shadow (same offsets as the flat buffer)
        char* sh = (dt == BLT_BF16) ? (char*)wshadow : nullptr;
        // a phase-1 step does not touch the latent-phase region: its shadow stays as valid as it was before this step
        const bool was_valid = shadow_gen == steps->params_gen;
        ++steps->params_gen;
        shadow_gen = (sh && (last_bwd_phase2 || was_valid)) ? steps->params_gen : -1;
        auto seg = [&](int64_t lo, int64_t hi, int step) -> int {
            if (hi <= lo) return BLT_OK;
            return blt_adam_step(train + lo, grad + lo, adam_m + lo, adam_v + lo, hi - lo, stats + 4, max_norm, lr, b1, b2, eps, step, s,
                                 sh ? sh + (size_t)lo * 2 : nullptr);
        };
        // stage 1: what the next forward reads first — context encoder, shared embedding, CNN head | posterior encoder
        RC(seg(dec_end, late_off, step_main));
        if (last_bwd_phase2) RC(seg(renc_off, tsize, step_late));
        if (staged && hipEventRecord(opt_stage1, s) != hipSuccess) { blt_set_error("engine_optimizer_step: event record failed"); return BLT_ERR_HIP; }
        // stage 2: vocabulary projection, reconstructor, decoder | z_classifier, latent projection, latent nets
        RC(seg(0, dec_end, step_main));
        if (last_bwd_phase2) RC(seg(late_off, renc_off, step_late));
        return BLT_OK;
    }
#undef RC
};

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" {

bltvqg_engine* bltvqg_engine_create(const bltvqg_config* cfg) {
    if (!cfg) { blt_set_error("engine_create: null config"); return nullptr; }
    const bltvqg_config& c = *cfg;
    if (c.batch <= 0 || c.hidden_dim <= 0 || c.hidden_dim % 8 != 0 || c.num_heads <= 0 || c.hidden_dim % c.num_heads != 0 ||
        c.pwffn_dim % 8 != 0 || c.latent_dim % 8 != 0 || c.emb_dim <= 0 || c.num_layers <= 0 || c.vocab_size < 6 ||
        c.len_context <= 0 || c.len_context > 64 || c.len_posterior <= 0 || c.len_posterior > 64 || c.len_target < 2 || c.len_target > 64 ||
        (c.num_regions <= 0 && (c.image_h < 32 || c.image_w < 32)) || c.num_regions < 0 ||
        (c.num_regions > 0 && (c.region_dim < 8 || c.region_dim % 8 != 0)) || c.region_pool < 0 || c.region_pool > 1 ||
        (c.region_pool == 1 && (c.num_regions <= 0 || c.num_regions > 64)) || (c.dtype != BLT_F32 && c.dtype != BLT_BF16) || c.hidden_dim > 2048 ||
        c.attention_dropout < 0.f || c.attention_dropout >= 1.f || c.relu_dropout < 0.f || c.relu_dropout >= 1.f) {
        blt_set_error("engine_create: unsupported configuration (need H,F,Z %% 8 == 0, H %% heads == 0, H <= 2048, sequence lengths <= 64, images >= 32x32 or region_dim %% 8 == 0)");
        return nullptr;
    }
    if (c.emb_dim % 4 != 0) { blt_set_error("engine_create: emb_dim must be a multiple of 4"); return nullptr; }
    if (c.head_dim_true < 0 || c.head_dim_true > c.hidden_dim / c.num_heads || (c.head_dim_true > 0 && (c.num_heads * c.head_dim_true) % 2 != 0)) {
        blt_set_error("engine_create: head_dim_true must be in [0, hidden_dim / num_heads] with an even true width");
        return nullptr;
    }
    return new bltvqg_engine(c);
}

void bltvqg_engine_destroy(bltvqg_engine* e) {
    if (!e) return;
    for (auto& b : e->buckets) if (b.ev) (void)hipEventDestroy(b.ev);
    if (e->grad_zero_ev) (void)hipEventDestroy(e->grad_zero_ev);
    for (int i = 0; i < 16; ++i) if (e->fj[i]) (void)hipEventDestroy(e->fj[i]);
    for (int i = 0; i < 2; ++i) if (e->side[i]) (void)hipStreamDestroy(e->side[i]);
    if (e->cnn_stream && e->cnn_stream_owned) (void)hipStreamDestroy(e->cnn_stream);
    if (e->chain_stream) (void)hipStreamDestroy(e->chain_stream);
    if (e->cnn_in_ev) (void)hipEventDestroy(e->cnn_in_ev);
    for (int i = 0; i < 2; ++i) if (e->cnn_done[i]) (void)hipEventDestroy(e->cnn_done[i]);
    if (e->opt_fork) (void)hipEventDestroy(e->opt_fork);
    if (e->opt_done) {
        if (e->steps->opt_done_ev == e->opt_done) {      // an update of this engine may still be in flight for the engines sharing its state
            (void)hipEventSynchronize(e->opt_done);
            e->steps->opt_done_ev = nullptr;
            e->steps->opt_stage1_ev = nullptr;
        }
        (void)hipEventDestroy(e->opt_done);
    }
    if (e->opt_stage1) (void)hipEventDestroy(e->opt_stage1);
    for (size_t i = 0; i < e->prof.size(); ++i) { (void)hipEventDestroy(e->prof[i].a); (void)hipEventDestroy(e->prof[i].b); }
    delete e;
}

int bltvqg_engine_num_params(const bltvqg_engine* e, int which) { return e ? (int)(which == 0 ? e->tp.size() : e->fp.size()) : 0; }

int bltvqg_engine_param_info(const bltvqg_engine* e, int which, int index, char* name_host, int name_cap, int64_t* offset, int64_t* numel,
                             int32_t* dims4_host, int32_t* ndim, int32_t* late) {
    BLT_REQUIRE(e && (which == 0 || which == 1), "engine_param_info: bad args");
    const std::vector<PInfo>& v = which == 0 ? e->tp : e->fp;
    BLT_REQUIRE(index >= 0 && index < (int)v.size(), "engine_param_info: index %d out of range", index);
    const PInfo& p = v[index];
    if (name_host && name_cap > 0) snprintf(name_host, name_cap, "%s", p.name.c_str());
    if (offset) *offset = p.off;
    if (numel) *numel = p.numel;
    if (dims4_host) for (int i = 0; i < 4; ++i) dims4_host[i] = p.dims[i];
    if (ndim) *ndim = p.ndim;
    if (late) *late = p.late;
    return BLT_OK;
}

int64_t bltvqg_engine_flat_size(const bltvqg_engine* e, int which) { return e ? (which == 0 ? e->tsize : e->fsize) : 0; }
int64_t bltvqg_engine_late_offset(const bltvqg_engine* e) { return e ? e->late_off : 0; }
int64_t bltvqg_engine_workspace_bytes(const bltvqg_engine* e) { return e ? e->ws_bytes : 0; }

int bltvqg_engine_bind(bltvqg_engine* e, float* train, float* grad, float* adam_m, float* adam_v, float* frozen, void* workspace,
                       int64_t workspace_bytes) {
    BLT_REQUIRE(e && train && grad && frozen && workspace, "engine_bind: null pointer");
    BLT_REQUIRE(workspace_bytes >= e->ws_bytes, "engine_bind: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)e->ws_bytes);
    BLT_REQUIRE(((uintptr_t)workspace % 256) == 0 && ((uintptr_t)train % 16) == 0 && ((uintptr_t)grad % 16) == 0 && ((uintptr_t)frozen % 16) == 0,
                "engine_bind: misaligned buffer");
    e->train = train; e->grad = grad; e->adam_m = adam_m; e->adam_v = adam_v; e->frozen = frozen; e->ws = (char*)workspace;
    e->layout(e->ws);
    {
        const hipError_t rc_ = hipMemset(workspace, 0, (size_t)e->ws_bytes);
        if (rc_ != hipSuccess) {
            blt_set_error("engine_bind: workspace memset of %lld bytes failed (%s)", (long long)e->ws_bytes, hipGetErrorString(rc_));
            return BLT_ERR_HIP;
        }
    }
    // sinusoid timing signal, transformer_layers.py:542-558: [sin(pos*w_i) | cos(pos*w_i)], float64 then cast
    {
        // (padded widths: the signal is that of the TRUE width, each real feature i stored at its padded column)
        const int H = e->H, Ht = e->H_true, nt = Ht / 2, dht = e->dh_true, dhp = e->dh;
        auto col = [&](int i) { return (i / dht) * dhp + (i % dht); };
        std::vector<float> tab((size_t)64 * H, 0.f);
        const double inc = log(1.0e4 / 1.0) / ((double)nt - 1.0);
        for (int pos = 0; pos < 64; ++pos)
            for (int i = 0; i < nt; ++i) {
                const double st = (double)pos * (1.0 * exp((double)i * -inc));
                tab[(size_t)pos * H + col(i)] = (float)sin(st);
                tab[(size_t)pos * H + col(nt + i)] = (float)cos(st);
            }
        if (hipMemcpy(e->timing, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            blt_set_error("engine_bind: timing table upload failed");
            return BLT_ERR_HIP;
        }
        if (e->ttable && !e->tlist.empty() &&
            hipMemcpy(e->ttable, e->tlist.data(), e->tlist.size() * sizeof(bltvqg_engine::TEnt), hipMemcpyHostToDevice) != hipSuccess) {
            blt_set_error("engine_bind: transposed-shadow table upload failed");
            return BLT_ERR_HIP;
        }
        for (int w = 0; w < 2; ++w)
            if (e->fold_ok && !e->fold_tab[w].empty() &&
                hipMemcpy(e->fold_tab_dev[w], e->fold_tab[w].data(), e->fold_tab[w].size() * sizeof(BltFoldEnt), hipMemcpyHostToDevice) != hipSuccess) {
                blt_set_error("engine_bind: LayerNorm-fold table upload failed");
                return BLT_ERR_HIP;
            }
    }
    if (!e->grad_zero_ev && hipEventCreateWithFlags(&e->grad_zero_ev, hipEventDisableTiming) != hipSuccess) {
        blt_set_error("engine_bind: event creation failed");
        return BLT_ERR_HIP;
    }
    for (auto& b : e->buckets)
        if (!b.ev && hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) {
            blt_set_error("engine_bind: event creation failed");
            return BLT_ERR_HIP;
        }
    for (int i = 0; i < 16; ++i)
        if (!e->fj[i] && hipEventCreateWithFlags(&e->fj[i], hipEventDisableTiming) != hipSuccess) {
            blt_set_error("engine_bind: event creation failed");
            return BLT_ERR_HIP;
        }
    for (int i = 0; i < 2; ++i)
        // (stream priorities were measured — either side stream lowest or highest: +-1 %, the dispatcher does not preempt resident workgroups)
        if (!e->side[i]) { const int rc_ = e->make_stream(&e->side[i], i); if (rc_) return rc_; }
    e->opt_stream = e->side[1];
    if (!e->cnn_in_ev && hipEventCreateWithFlags(&e->cnn_in_ev, hipEventDisableTiming) != hipSuccess) { blt_set_error("engine_bind: event creation failed"); return BLT_ERR_HIP; }
    for (int i = 0; i < 2; ++i)
        if (!e->cnn_done[i] && hipEventCreateWithFlags(&e->cnn_done[i], hipEventDisableTiming) != hipSuccess) {
            blt_set_error("engine_bind: event creation failed");
            return BLT_ERR_HIP;
        }
    e->pf_n = 0; e->pool_in_use = false;
    if ((!e->opt_fork && hipEventCreateWithFlags(&e->opt_fork, hipEventDisableTiming) != hipSuccess) ||
        (!e->opt_stage1 && hipEventCreateWithFlags(&e->opt_stage1, hipEventDisableTiming) != hipSuccess) ||
        (!e->opt_done && hipEventCreateWithFlags(&e->opt_done, hipEventDisableTiming) != hipSuccess)) {
        blt_set_error("engine_bind: event creation failed");
        return BLT_ERR_HIP;
    }
    e->bound = true; e->frozen_dirty = true; e->fwd_done = false; e->opt_mark_synced();
    return BLT_OK;
}

void bltvqg_engine_invalidate_frozen(bltvqg_engine* e) {
    if (!e) return;
    e->frozen_dirty = true;
    ++e->steps->params_gen;        // parameters were (re)written from outside: every engine sharing them rebuilds its weight shadows
}
void bltvqg_engine_invalidate_params(bltvqg_engine* e) {
    if (e) ++e->steps->params_gen;      // trainable parameters may have been written from outside (a torch optimiser): no shadow is current
}
int bltvqg_engine_trust_shadows(bltvqg_engine* e, int on) {
    BLT_REQUIRE(e, "engine_trust_shadows: null engine");
    e->trust_shadows = on != 0;
    return BLT_OK;
}

int bltvqg_engine_forward(bltvqg_engine* e, const float* images, const int64_t* context, const int64_t* posterior, const int64_t* target,
                          const float* eps, int phase2, uint64_t seed, void* stream) {
    BLT_REQUIRE(e, "engine_forward: null engine");
    return e->forward(images, context, posterior, target, eps, phase2, seed, (hipStream_t)stream);
}

int bltvqg_engine_prefetch_images(bltvqg_engine* e, const float* images, void* stream) {
    BLT_REQUIRE(e, "engine_prefetch_images: null engine");
    return e->prefetch_images(images, (hipStream_t)stream);
}
int bltvqg_engine_prefetch_pending(const bltvqg_engine* e) { return e ? e->pf_n : 0; }
int bltvqg_engine_set_prefetch_split(bltvqg_engine* e, int stages) {
    BLT_REQUIRE(e && stages >= 1 && stages <= bltvqg_engine::CNN_STAGES, "engine_set_prefetch_split: stages must be 1..10");
    BLT_REQUIRE(e->pf_n == 0, "engine_set_prefetch_split: a prefetched batch is pending");
    e->prefetch_split = stages;
    return BLT_OK;
}

int bltvqg_engine_set_cu_masks(bltvqg_engine* e, const uint32_t* chain_mask_host, const uint32_t* side_mask_host, const uint32_t* conv_mask_host,
                               int words, int chain_cus) {
    BLT_REQUIRE(e && words >= 1 && words <= 8 && chain_cus >= 0 && chain_cus <= 1024, "engine_set_cu_masks: bad args");
    BLT_REQUIRE(e->pf_n == 0 && !e->opt_is_pending(), "engine_set_cu_masks: a prefetched batch or an optimiser update is pending");
    const uint32_t* src[3] = {chain_mask_host, side_mask_host, conv_mask_host};
    e->have_masks = false;
    for (int k = 0; k < 3; ++k) {
        e->cu_mask_on[k] = src[k] != nullptr;
        for (int w = 0; w < 8; ++w) e->cu_masks[k][w] = (src[k] && w < words) ? src[k][w] : 0u;
        if (src[k]) {
            uint32_t any = 0;
            for (int w = 0; w < words; ++w) any |= src[k][w];
            BLT_REQUIRE(any != 0, "engine_set_cu_masks: empty mask %d", k);
            e->have_masks = true;
        }
    }
    // every stream the engine owns is recreated under its mask (the caller has synchronised: nothing of this engine is in flight)
    if (hipDeviceSynchronize() != hipSuccess) { blt_set_error("engine_set_cu_masks: synchronize failed"); return BLT_ERR_HIP; }
    for (int i = 0; i < 2; ++i) if (e->side[i]) { (void)hipStreamDestroy(e->side[i]); e->side[i] = nullptr; }
    if (e->cnn_stream && e->cnn_stream_owned) { (void)hipStreamDestroy(e->cnn_stream); e->cnn_stream = nullptr; }
    if (e->chain_stream) { (void)hipStreamDestroy(e->chain_stream); e->chain_stream = nullptr; }
    if (e->bound) {
        { const int rc_ = e->make_stream(&e->side[0], 0); if (rc_) return rc_; }
        { const int rc_ = e->make_stream(&e->side[1], 1); if (rc_) return rc_; }
        e->opt_stream = e->side[1];
    }
    e->cnn_stream_used = false;
    blt_set_plan_cus(chain_cus > 0 ? chain_cus : 0);
    return BLT_OK;
}
int bltvqg_engine_chain_stream(bltvqg_engine* e, void** stream) {
    BLT_REQUIRE(e && stream, "engine_chain_stream: null argument");
    if (!e->chain_stream) { const int rc_ = e->make_stream(&e->chain_stream, 0); if (rc_) return rc_; }
    *stream = (void*)e->chain_stream;
    return BLT_OK;
}

int bltvqg_engine_adopt_conv_stream(bltvqg_engine* e, void* stream) {
    BLT_REQUIRE(e && stream, "engine_adopt_conv_stream: null argument");
    BLT_REQUIRE(e->pf_n == 0, "engine_adopt_conv_stream: a prefetched batch is pending");
    if (e->cnn_stream && e->cnn_stream_owned) { (void)hipStreamSynchronize(e->cnn_stream); (void)hipStreamDestroy(e->cnn_stream); }
    e->cnn_stream = (hipStream_t)stream;
    e->cnn_stream_owned = false;
    return BLT_OK;
}
int bltvqg_engine_conv_stream(bltvqg_engine* e, void** stream) {
    BLT_REQUIRE(e && stream, "engine_conv_stream: null argument");
    if (!e->cnn_stream) { const int rc_ = e->make_stream(&e->cnn_stream, 2); if (rc_) return rc_; }
    *stream = (void*)e->cnn_stream;
    return BLT_OK;
}

int bltvqg_engine_side_stream(bltvqg_engine* e, int which, void** stream) {
    BLT_REQUIRE(e && stream && (which == 0 || which == 1), "engine_side_stream: bad args");
    BLT_REQUIRE(e->bound && e->side[which], "engine_side_stream: engine not bound");
    *stream = (void*)e->side[which];
    return BLT_OK;
}

int bltvqg_engine_conv_stream_wait(bltvqg_engine* e, void* stream) {
    BLT_REQUIRE(e, "engine_conv_stream_wait: null engine");
    if (!e->cnn_stream_used || !e->cnn_done[e->last_cnn_slot]) return BLT_OK;      // nothing ran ahead
    if (hipStreamWaitEvent((hipStream_t)stream, e->cnn_done[e->last_cnn_slot], 0) != hipSuccess) { blt_set_error("engine_conv_stream_wait: wait failed"); return BLT_ERR_HIP; }
    return BLT_OK;
}

int bltvqg_engine_image_input(bltvqg_engine* e, void** ptr, int* Hp, int* Wp, int* dtype) {
    BLT_REQUIRE(e && ptr && Hp && Wp && dtype, "engine_image_input: null argument");
    BLT_REQUIRE(e->bound && !e->regions, "engine_image_input: engine not bound / region mode has no image input");
    *ptr = e->img; *Hp = e->imgHp; *Wp = e->imgWp; *dtype = e->dt;
    return BLT_OK;
}

int bltvqg_engine_decode_greedy(bltvqg_engine* e, const float* images, const int64_t* context, const float* eps, int phase2, int train_bn,
                                int32_t* tokens, int32_t* top_idx, float* top_val, void* stream) {
    BLT_REQUIRE(e, "engine_decode_greedy: null engine");
    return e->decode_greedy(images, context, eps, phase2, train_bn, tokens, top_idx, top_val, (hipStream_t)stream);
}

int bltvqg_engine_set_bn_train(bltvqg_engine* e, int train) {
    BLT_REQUIRE(e, "engine_set_bn_train: null engine");
    e->bn_train = train != 0;
    return BLT_OK;
}

int bltvqg_engine_loss_backward(bltvqg_engine* e, float kl_weight, void* stream) {
    BLT_REQUIRE(e, "engine_loss_backward: null engine");
    return e->loss_backward(kl_weight, (hipStream_t)stream);
}

int bltvqg_engine_backward_external(bltvqg_engine* e, const float* d_output, const float* d_zlogit, float d_kld, const float* d_feats,
                                    const float* d_recon, void* stream) {
    BLT_REQUIRE(e, "engine_backward_external: null engine");
    return e->backward_external(d_output, d_zlogit, d_kld, d_feats, d_recon, (hipStream_t)stream);
}

int bltvqg_engine_optimizer_step(bltvqg_engine* e, float lr, float max_norm, float beta1, float beta2, float eps, void* stream) {
    BLT_REQUIRE(e, "engine_optimizer_step: null engine");
    return e->optimizer_step(lr, max_norm, beta1, beta2, eps, (hipStream_t)stream);
}
int bltvqg_engine_optimizer_step_async(bltvqg_engine* e, float lr, float max_norm, float beta1, float beta2, float eps, void* stream) {
    BLT_REQUIRE(e, "engine_optimizer_step_async: null engine");
    return e->optimizer_step(lr, max_norm, beta1, beta2, eps, (hipStream_t)stream, true);
}
int bltvqg_engine_adam_steps(const bltvqg_engine* e, int32_t* steps_main_host, int32_t* steps_late_host) {
    BLT_REQUIRE(e && steps_main_host && steps_late_host, "engine_adam_steps: bad args");
    *steps_main_host = e->steps->main;
    *steps_late_host = e->steps->late;
    return BLT_OK;
}
int bltvqg_engine_set_adam_steps(bltvqg_engine* e, int32_t steps_main, int32_t steps_late) {
    BLT_REQUIRE(e && steps_main >= 0 && steps_late >= 0, "engine_set_adam_steps: bad args");
    e->steps->main = steps_main;
    e->steps->late = steps_late;
    return BLT_OK;
}
int bltvqg_engine_share_optimizer_state(bltvqg_engine* e, bltvqg_engine* primary) {
    BLT_REQUIRE(e && primary && e->tsize == primary->tsize, "engine_share_optimizer_state: engines of different models");
    e->steps = primary->steps;
    e->opt_seen = -1;      // unknown: wait for whatever update of the primary may be in flight before touching the shared buffers
    if (e->steps->opt_gen == -1) e->opt_seen = -2;
    return BLT_OK;
}
int bltvqg_engine_optimizer_wait(bltvqg_engine* e, void* stream) {
    BLT_REQUIRE(e, "engine_optimizer_wait: null engine");
    if (!e->opt_is_pending()) return BLT_OK;
    const int rc = e->sync_opt((hipStream_t)stream);
    e->opt_mark_synced();
    return rc;
}

int bltvqg_engine_read(bltvqg_engine* e, int what, float* dst, void* stream) {
    BLT_REQUIRE(e && e->bound && dst, "engine_read: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (e->opt_is_pending()) { const int rc_ = e->sync_opt(s); if (rc_) return rc_; }     // stats[4] / parameters of a pending update
    switch (what) {
        case 0: return blt_cast_rows(e->dt, e->logits, e->ldV, BLT_F32, dst, e->V, e->Mt, e->V, s);
        case 1: return blt_cast_rows(e->dt, e->zlogit, e->ldV, BLT_F32, dst, e->V, e->B, e->V, s);
        case 2: return blt_cast_rows(BLT_F32, e->feats32, e->H, BLT_F32, dst, e->H, e->B, e->H, s);
        case 3: return blt_cast_rows(e->dt, e->recon, e->H, BLT_F32, dst, e->H, e->B, e->H, s);
        case 4:
            if (hipMemcpyAsync(dst, e->stats, 8 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return BLT_ERR_HIP;
            if (hipMemcpyAsync(dst + 5, e->counters, sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return BLT_ERR_HIP;
            return hipMemcpyAsync(dst + 6, e->stats - 1, sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess ? BLT_OK : BLT_ERR_HIP;
        case 5: return blt_cast_rows(e->dt, e->enc.out, e->H, BLT_F32, dst, e->H, e->Ma, e->H, s);
        case 6: return blt_cast_rows(e->dt, e->dec.out, e->H, BLT_F32, dst, e->H, e->Mt, e->H, s);
        default:
            if (what >= 100 && what < 100 + (int)e->convs.size()) {      // debug: conv stage outputs (NHWC), after their in-place BatchNorm
                const ConvSpec& cs = e->convs[what - 100];
                return blt_cast_rows(e->dt, cs.out, cs.Cout, BLT_F32, dst, cs.Cout, (long)e->B * cs.Ho * cs.Wo, cs.Cout, s);
            }
            blt_set_error("engine_read: unknown item %d", what); return BLT_ERR_ARG;
    }
}

uint32_t bltvqg_engine_dropout_stream_id(int stack, int layer, int site) { return (uint32_t)(stack * 1000 + layer * 10 + site); }

int bltvqg_engine_profile_enable(bltvqg_engine* e, int mask) {
    BLT_REQUIRE(e && mask >= 0 && (mask & 0xFF) <= 3 && (mask >> 8) <= 64, "engine_profile_enable: bad args");
    e->prof_mask = mask & 3;    // pause / resume: the recorded launches accumulate until bltvqg_engine_profile_read* drains them
    e->prof_stride = (mask >> 8) > 0 ? (mask >> 8) : 1;      // bits 8..: sample every n-th plain Linear GEMM launch (weighted n)
    e->prof_gemm_count = 0;
    return BLT_OK;
}

int bltvqg_engine_profile_read_class(bltvqg_engine* e, int cls, double* total_ms_host, int32_t* launches_host, double* flops_host) {
    return bltvqg_engine_profile_read_streams(e, cls, total_ms_host, launches_host, flops_host, nullptr);
}
int bltvqg_engine_profile_read_streams(bltvqg_engine* e, int cls, double* total_ms_host, int32_t* launches_host, double* flops_host, double* per_stream_ms_host4) {
    BLT_REQUIRE(e && (cls == 0 || cls == 1) && total_ms_host && launches_host && flops_host, "engine_profile_read: bad args");
    double total = 0.0, flops = 0.0, by_stream[4] = {0.0, 0.0, 0.0, 0.0};
    int32_t n = 0;
    size_t keep = 0;
    for (size_t i = 0; i < e->prof_n; ++i) {
        bltvqg_engine::ProfRec& r = e->prof[i];
        if (r.cls != cls) { std::swap(e->prof[keep], e->prof[i]); ++keep; continue; }
        if (hipEventSynchronize(r.b) != hipSuccess) { blt_set_error("engine_profile_read: event sync failed"); return BLT_ERR_HIP; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { blt_set_error("engine_profile_read: elapsed failed"); return BLT_ERR_HIP; }
        total += (double)ms * r.w; flops += r.flops * r.w; n += r.w;
        by_stream[r.sidx & 3] += (double)ms * r.w;
    }
    if (per_stream_ms_host4) for (int k = 0; k < 4; ++k) per_stream_ms_host4[k] = by_stream[k];
    e->prof_n = keep;      // records of the other class stay queued (the swaps keep every event pair alive in the vector)
    *total_ms_host = total;
    *launches_host = n;
    *flops_host = flops;
    return BLT_OK;
}
int bltvqg_engine_profile_read(bltvqg_engine* e, double* total_ms_host, int32_t* launches_host, double* flops_host) {
    return bltvqg_engine_profile_read_class(e, 0, total_ms_host, launches_host, flops_host);
}

int bltvqg_engine_phase_stamps(bltvqg_engine* e, float* ms_host12) {
    BLT_REQUIRE(e && ms_host12, "engine_phase_stamps: bad args");
    for (int i = 0; i < 12; ++i) ms_host12[i] = -1.f;
    if (!e->stamp_used[0]) return BLT_OK;
    for (int i = 1; i < 12; ++i) {
        if (!e->stamp_used[i]) continue;
        if (hipEventSynchronize(e->stamp_ev[i]) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->stamp_ev[0], e->stamp_ev[i]) == hipSuccess) ms_host12[i] = ms;
    }
    ms_host12[0] = 0.f;
    return BLT_OK;
}

int bltvqg_engine_set_bucket_flush(bltvqg_engine* e, int on) {
    BLT_REQUIRE(e, "engine_set_bucket_flush: null engine");
    e->group_flush = on != 0;
    return BLT_OK;
}
int bltvqg_engine_num_buckets(const bltvqg_engine* e) { return e ? (int)e->buckets.size() : 0; }
int bltvqg_engine_bucket_info(const bltvqg_engine* e, int i, int64_t* offset, int64_t* numel, int32_t* late) {
    BLT_REQUIRE(e && i >= 0 && i < (int)e->buckets.size(), "engine_bucket_info: bad args");
    if (offset) *offset = e->buckets[i].off;
    if (numel) *numel = e->buckets[i].len;
    if (late) *late = e->buckets[i].late;
    return BLT_OK;
}
int bltvqg_engine_bucket_wait(bltvqg_engine* e, int i, void* stream) {
    BLT_REQUIRE(e && i >= 0 && i < (int)e->buckets.size() && e->buckets[i].ev, "engine_bucket_wait: bad args");
    return hipStreamWaitEvent((hipStream_t)stream, e->buckets[i].ev, 0) == hipSuccess ? BLT_OK : BLT_ERR_HIP;
}

}  // extern "C"
