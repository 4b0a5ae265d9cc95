#!/usr/bin/env python3
"""Headline benchmark: image-question pairs/s of the BLT-VQG IQ train step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one full reference training step (train_iq.py:105-132 + Lightning's backward / clip 5 / Adam) on one synthetic
minibatch per GPU, inputs already resident in HBM: frozen ResNet-18 forward (train-mode BatchNorm), both transformer
encoders, latent, decoder, losses, the whole backward, gradient all-reduce (N > 1), clip + Adam.  Dropout is on at the
reference's 0.1/0.1.  Workload = BASELINE.json configs[2] (the configuration the metric's targets are quoted on): the 6-layer
d_model=512 8-head model at per-GPU batch 256, bf16; `--config small|regions` select configs[1] / configs[4].

N > 1: one process per GPU over RCCL.  Started under `torch.distributed.run` (WORLD_SIZE in the environment) this process is one
rank; started plainly with `--gpus N` it spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD before
any GPU call (the parent never initialises the GPU) and relays rank 0's JSON line — the equivalent of the reference's one flag
`pl.Trainer(gpus=N)` (train_iq.py:372-373).  Prints ONE JSON line on stdout.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP multiplexes streams onto 4 hardware queues by default; the step uses the main stream, two engine side streams and (N > 1) the
# trainer's communication stream plus RCCL's own: with 8 queues none of them is silently serialised behind another
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

CONFIGS = {
    # BASELINE.json configs[0]/[1]: train_iq.py defaults-shaped model (small)
    "small": dict(hidden_dim=256, pwffn_dim=512, latent_dim=256, emb_dim=300, num_layers=2, num_heads=4, vocab_size=8000, batch=128),
    # BASELINE.json configs[2]/[3]: 6-layer d_model=512 8-head (big)
    "big": dict(hidden_dim=512, pwffn_dim=2048, latent_dim=512, emb_dim=300, num_layers=6, num_heads=8, vocab_size=8000, batch=256),
    # BASELINE.json configs[4]: bottom-up features (36 x 2048 regions, no CNN), 6-layer transformer, global batch 512 = 8 x 64
    "regions": dict(hidden_dim=512, pwffn_dim=2048, latent_dim=512, emb_dim=300, num_layers=6, num_heads=8, vocab_size=8000, batch=64,
                    num_regions=36, region_dim=2048),
    # the one launch the reference documents (run.sh:1-10): hidden / latent 1024, FFN 2048, 6 layers, 8 heads of 128, batch 64, --input_mode cat
    "runsh": dict(hidden_dim=1024, pwffn_dim=2048, latent_dim=1024, emb_dim=300, num_layers=6, num_heads=8, vocab_size=8000, batch=64, len_context=3),
    # the reference's CLI defaults (train_iq.py:315-339): hidden 300 = 4 heads of 75, latent 300, FFN 600, 4 layers, batch 128 — on the padded
    # engine layout (heads in slots of 80 columns: hidden 320, latent 304; blt-vqg_amd/padded.py)
    "default300": dict(hidden_dim=300, pwffn_dim=600, latent_dim=300, emb_dim=300, num_layers=4, num_heads=4, vocab_size=8000, batch=128),
}
CONFIG_NAMES = {"small": "BASELINE configs[1]", "big": "BASELINE configs[2]", "regions": "BASELINE configs[4], one GPU's shard",
                "runsh": "the reference's documented launch (run.sh:1-10)", "default300": "the reference's CLI defaults (train_iq.py:315-339)"}
# algorithmic FLOP per pair per train step: CNN fwd x1 + everything trainable x3; regions: the projection runs on the
# region mean (the mean commutes with the Linear), 2.1 MFLOP per pair instead of the survey's 75.5
FLOP_PER_PAIR = {"small": 4.25e9, "big": 9.95e9, "regions": 6.33e9, "runsh": 20.0e9, "default300": 4.89e9}      # (SURVEY §8d formulas)
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HP = dict(kl_weight=0.5, lr=1e-4, max_norm=5.0)


PROFILE_STRIDE = 4      # the profiled step brackets every 4th plain Linear GEMM launch (weighted 4), see engine_profile_enable

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="big", choices=sorted(CONFIGS))
    ap.add_argument("--h2d", action="store_true",
                    help="make the PCIe-inclusive variant the timed loop (NOT the headline value): the batch lives in pinned host memory and "
                         "is copied to the GPU every step on a copy stream, one step ahead (the reference's boundary hands over host tensors, "
                         "train_iq.py:67-79).  Without this flag that variant is still measured, as the extra field `h2d`")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--phase", type=int, default=2, choices=[1, 2], help="timed loop: 1 = pre-training (latent off), 2 = latent on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra fields (phase-1 / PCIe-inclusive loops, per-bucket all-reduce timing)")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--bf16-wire", action="store_true", help="N > 1: gradients cross xGMI as bf16 (half the bytes); default fp32 like Lightning DDP")
    ap.add_argument("--autotune", action="store_true", help="time candidate GEMM/conv kernels per launch in the first warm-up step")
    ap.add_argument("--f32-steps", type=int, default=10, help="extra leg: timed steps of the fp32 (reference-precision, exact-fp32 MFMA) engine on the same "
                                                              "config, reported as `f32` beside the bf16 headline (0 = skip)")
    ap.add_argument("--no-prefetch", action="store_true", help="run the frozen conv stack inline in forward (round-2 form) instead of one batch ahead")
    ap.add_argument("--comm-stream", default="shared", choices=["shared", "own"],
                    help="N > 1: 'shared' = ONE stream carries the conv look-ahead of batch i+1 and then the gradient all-reduces of step i (four live "
                         "streams); 'own' = the collectives and the conv look-ahead each get a stream (DataParallelStep(comm_stream=...); a fifth live stream: the "
                         "one-rank rehearsal measured 13.1 ms / step against 7.5 ms shared, DESIGN.md section 6)")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=20)
    return ap.parse_args()


def spawn_ranks(a):
    """`--gpus N` without a launcher: start N ranks as a child process group and relay rank 0's JSON line — the launcher train_iq.py's
    `--num_gpus N` uses (blt-vqg_amd/launch.py, loaded by file path: standard library only).  Nothing here touches the GPU (no
    torch.cuda call, no HIP library load): exec'ing or forking from a GPU-initialised process is not allowed on these boxes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("blt_launch", os.path.join(ROOT, "blt-vqg_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    return launch.spawn_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:], relay_prefix='{"metric"')


def region_features(B, R, D, seed):
    """Synthetic bottom-up features: non-negative (post-ReLU) with a per-sample component."""
    import torch
    g = torch.Generator().manual_seed(int(seed) + 77)
    x = torch.relu(torch.randn(B, R, D, generator=g) + 0.3 * torch.randn(B, 1, D, generator=g))
    return (x * (0.5 + torch.rand(B, 1, 1, generator=g))).contiguous()


def oracle_namespace(cfg):
    from types import SimpleNamespace
    return SimpleNamespace(emb_dim=cfg["emb_dim"], hidden_dim=cfg["hidden_dim"], latent_dim=cfg["latent_dim"], pwffn_dim=cfg["pwffn_dim"],
                           num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], vocab_size=cfg["vocab_size"],
                           num_regions=cfg.get("num_regions", 0), region_dim=cfg.get("region_dim", 0))


def host_threads():
    """Threads of the CPU leg: the cores this process may run on (its affinity mask: the GPU box gives a job a share of the host),
    at most 64 — beyond that the oracle's batch-32 step stops scaling and only oversubscribes.  Returns (threads, affinity, os count)."""
    import torch
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    threads = max(1, min(affinity, 64))
    torch.set_num_threads(threads)
    return threads, affinity, (os.cpu_count() or 1)


def thread_candidates(limit):
    """Thread counts the CPU leg tries (one untimed + one timed step each) before its measured loop: the oracle's batch-32 step does not
    scale to every core the box has (measured on a 256-CPU host: 0.46 s per step with 16 threads, 1.49 s with 64), and the baseline should be
    the CPU's best, not a thread count picked by rule."""
    return sorted({t for t in (8, 16, 32, limit) if 1 <= t <= limit})


def cpu_baseline(cfg, phase2, batch, steps):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference, pinned by tests/golden) timed on the host cores:
    full train step (forward, losses, backward, clip, Adam) on the same synthetic workload at a bounded batch."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import iq_oracle as O
    from synth import synth_state
    import bltvqg_amd.synthetic as synthetic
    ns = oracle_namespace(cfg)
    limit, affinity, os_count = host_threads()
    print("[bench] cpu baseline: up to %d threads (affinity %d, os.cpu_count() = %d)" % (limit, affinity, os_count), file=sys.stderr, flush=True)
    state = synth_state(O.iq_spec(ns), seed=1)
    P = O.clone_params(state)
    names = O.trainable_names(P)
    opt = torch.optim.Adam([P[n] for n in names], lr=1e-4)
    hp = O.default_hp()
    b = synthetic.make_batch(batch, ns.vocab_size, ns.latent_dim, seed=1234, image_hw=32 if ns.num_regions else 224)
    if cfg.get("len_context", 5) == 3:      # --input_mode cat
        b["answers"] = b["answer_types_for_input"]
    if ns.num_regions:
        b["images"] = region_features(batch, ns.num_regions, ns.region_dim, 1234)
    gen = torch.Generator().manual_seed(7)

    def one():
        out, z_logit, kld, recon, _ = O.iq_forward(P, ns, b["images"], b["answers"], b["posteriors"], b["questions"], phase2,
                                                   torch.randn(batch, ns.latent_dim, generator=gen), None, 0.0, True, {})
        loss, _ = O.calculate_losses(out, recon, kld, z_logit, b["questions"], phase2, 100, hp)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([P[n] for n in names if P[n].grad is not None], 5.0)
        opt.step()
    tried = {}
    for t in thread_candidates(limit):
        torch.set_num_threads(t)
        one()
        t0 = time.perf_counter()
        one()
        tried[t] = round(time.perf_counter() - t0, 3)
        print("[bench] cpu baseline: %d threads -> %.2f s per step" % (t, tried[t]), file=sys.stderr, flush=True)
    cores = min(tried, key=tried.get)
    torch.set_num_threads(cores)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        out, z_logit, kld, recon, _ = O.iq_forward(P, ns, b["images"], b["answers"], b["posteriors"], b["questions"], phase2,
                                                   torch.randn(batch, ns.latent_dim, generator=gen), None, 0.0, True, {})
        loss, _ = O.calculate_losses(out, recon, kld, z_logit, b["questions"], phase2, 100, hp)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([P[n] for n in names if P[n].grad is not None], 5.0)
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
        print("[bench] cpu baseline step %d: %.2f s" % (i, time.perf_counter() - t0), file=sys.stderr, flush=True)
    t = sorted(times)[len(times) // 2]
    return dict(value=round(batch / t, 2), unit="pairs/s", cores=cores, threads=cores, threads_tried_s_per_step={str(k): v for k, v in tried.items()},
                affinity_cpus=affinity, os_cpu_count=os_count, kind="port",
                s_per_step=round(t, 3), batch=batch,
                sample="%d timed train steps (median, 1 untimed warm-up) of the CPU oracle: same model config and synthetic inputs at batch %d, "
                       "fp32, phase %d (latent %s), dropout OFF (the GPU leg runs the reference's 0.1/0.1: Philox masks cost the CPU leg nothing "
                       "it would not also skip), forward + losses + backward + clip 5 + Adam" % (steps, batch, 2 if phase2 else 1,
                                                                                                 "on" if phase2 else "off"))


def loss_vs_oracle(eng, cfg, a, B, phase2, batch, rank, engine_config=None, eZ=None):
    """One dropout-off step of the BENCHED dtype / config / batch on the GPU (an engine of the same shape with dropout 0 that shares the
    timed engine's current parameters) against the CPU oracle's forward + losses on the same inputs and parameters."""
    import torch
    from bltvqg_amd.engine import StepEngine, make_config
    from oracle import iq_oracle as O
    dev = eng.device
    c = engine_config(1 if a.dtype == "bf16" else 0, attention_dropout=0.0, relu_dropout=0.0)
    e0 = StepEngine(c, dev)
    e0.allocate(share_from=eng)
    ns = oracle_namespace(cfg)
    state = {n: eng.view(n, 0).detach().cpu().clone() for n in eng.train_info}
    state.update({n: eng.view(n, 1).detach().cpu().clone() for n in eng.frozen_info})
    for k, shape in O.iq_spec(ns).items():
        if k not in state:      # num_batches_tracked counters
            state[k] = torch.zeros(shape, dtype=torch.long)
    eps = batch["eps"]
    d = {k: batch[k].to(dev) for k in ("images", "answers", "posteriors", "questions")}
    e0.forward(d["images"], d["answers"], d["posteriors"], d["questions"], eps.to(dev) if phase2 else None, phase2, 0)
    e0.loss_backward(HP["kl_weight"])
    st = e0.stats()
    gpu = dict(rec=st["rec"], img=st["img"], kld=st["kld"] if phase2 else 0.0, aux=st["aux"] if phase2 else 0.0)
    t0 = time.perf_counter()
    with torch.no_grad():
        out, z_logit, kld, recon, _ = O.iq_forward(O.clone_params(state, requires_grad=False), ns, batch["images"], batch["answers"],
                                                   batch["posteriors"], batch["questions"], phase2, eps, None, 0.0, True, {})
        hp = O.default_hp()
        _, ost = O.calculate_losses(out, recon, kld, z_logit, batch["questions"], phase2, 100, hp)
    print("[bench] oracle forward + losses at batch %d: %.1f s" % (B, time.perf_counter() - t0), file=sys.stderr, flush=True)
    ref = dict(rec=ost["rec"], img=ost["img"], kld=ost["kld"], aux=ost["aux"])

    def total(x):
        return x["rec"] + hp.image_recon_lambda * x["img"] + (hp.kl_ceiling * HP["kl_weight"] * x["kld"] + hp.aux_ceiling * x["aux"] if phase2 else 0.0)
    tg, tr = total(gpu), total(ref)
    return {"gpu_loss": round(tg, 6), "oracle_loss": round(tr, 6), "abs": round(abs(tg - tr), 6), "rel": round(abs(tg - tr) / max(abs(tr), 1e-30), 6),
            "components_abs": {k: round(abs(gpu[k] - ref[k]), 6) for k in ref},
            "what": "total loss (rec + 0.1 img%s) of ONE dropout-off step, %s engine at batch %d with the timed engine's current parameters, injected eps, "
                    "vs the fp32 CPU oracle on the same inputs" % (" + 0.5*%.2f kld + aux" % HP["kl_weight"] if phase2 else "", a.dtype, B)}


WORK_SKIPPING_KEYS = (14, 15)


def check_debug_keys(lib):
    """{key: value} of every non-zero debug key; exits non-zero when a work-skipping ablation is requested or available."""
    keys = {str(k): int(lib.bltvqg_debug_get(k)) for k in range(32) if int(lib.bltvqg_debug_get(k)) != 0}
    bad = [k for k in WORK_SKIPPING_KEYS if str(k) in keys]
    if bad or int(lib.bltvqg_build_has_ablations()):
        print("bench.py: refusing to time a run that can skip work (debug keys %s set, ablation build: %d); the headline needs the shipped "
              "library and keys 14 / 15 unset" % (bad, int(lib.bltvqg_build_has_ablations())), file=sys.stderr, flush=True)
        sys.exit(3)
    return keys


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    import torch
    if os.environ.get("BLT_BENCH_REHEARSE_LAUNCH") == "1":
        # CPU rehearsal of the launch path only (tests/test_bench_launch.py): the ranks meet over gloo, prove that the collective spans
        # `--gpus` ranks and rank 0 prints a line of the same shape; nothing is measured and no GPU is touched
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({"metric": "launch rehearsal (no measurement)", "n_gpus": dist.get_world_size(), "ranks_seen": int(t.item())}), flush=True)
        dist.destroy_process_group()
        return
    # stdout carries exactly ONE line, the JSON: native libraries that write to file descriptor 1 (RCCL prints a version banner there
    # when NCCL_DEBUG=VERSION is set, as it is on the GPU boxes) are sent to stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        print("warning: --gpus %d but WORLD_SIZE %d (the launcher's world size is what runs)" % (a.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    # BLT_SHARE_GPU=1 + BLT_DIST_BACKEND=gloo: functional rehearsal of the N-rank step on a box with fewer GPUs than ranks (every rank on
    # its LOCAL_RANK modulo the device count, gloo moving the device-resident gradients); the line it prints says so and is no measurement
    share_gpu = os.environ.get("BLT_SHARE_GPU") == "1"
    if share_gpu:
        local = local % torch.cuda.device_count()
    backend = os.environ.get("BLT_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("BLT_FORCE_DIST") == "1":      # BLT_FORCE_DIST: rehearse the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world

    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import DataParallelStep, comm_plan, init_reference_style, shard_seed

    # A/B switches of the kernels (bltvqg_debug_set): BLT_DEBUG="key=value,key=value".  Every non-zero key is echoed in the JSON line
    # (`debug_keys`); the work-skipping timing ablations (keys 14 / 15, which exist only in the -DBLT_ABLATE build) are refused here: a
    # timed region that skips work is not a measurement of this benchmark.
    from bltvqg_amd import _lib as _l
    if os.environ.get("BLT_DEBUG"):
        for kv in os.environ["BLT_DEBUG"].split(","):
            k, v = kv.split("=")
            _l.load().bltvqg_debug_set(int(k), int(v))
    debug_keys = check_debug_keys(_l.load())
    cfg = dict(CONFIGS[a.config])
    B = a.batch or cfg.pop("batch")
    cfg.pop("batch", None)
    len_context = cfg.get("len_context", 5)
    # widths the kernels do not tile directly (the reference's default 300 = 4 heads of 75) run on the padded layout: the engine gets the
    # padded widths + the true head width, the pad positions of every parameter are zero and stay zero (blt-vqg_amd/padded.py)
    from bltvqg_amd.padded import PaddedLayout, needs_padding
    pad = PaddedLayout(cfg["hidden_dim"], cfg["latent_dim"], cfg["pwffn_dim"], cfg["num_heads"]) \
        if needs_padding(cfg["hidden_dim"], cfg["latent_dim"], cfg["pwffn_dim"], cfg["num_heads"]) else None
    eH, eF, eZ = (pad.Hp, pad.Fp, pad.Zp) if pad else (cfg["hidden_dim"], cfg["pwffn_dim"], cfg["latent_dim"])

    def engine_config(dtype, **kw):
        return make_config(B, eH, eF, eZ, cfg["emb_dim"], cfg["num_layers"], cfg["num_heads"], cfg["vocab_size"], len_context=len_context,
                           dtype=dtype, num_regions=cfg.get("num_regions", 0), region_dim=cfg.get("region_dim", 0),
                           head_dim_true=pad.dh if pad else 0, **kw)

    def init_engine(e):
        init_reference_style(e, seed=0)                    # same weights on every rank
        if pad:                                            # zeros at the pad positions
            for flat, infos in ((e.flat_train, e.train_info), (e.flat_frozen, e.frozen_info)):
                _, index, _ = pad.build(infos)
                keep = torch.zeros(flat.numel(), dtype=torch.bool, device=flat.device)
                keep[index.to(flat.device)] = True
                flat.mul_(keep)
            e.params_changed()
    c = engine_config(1 if a.dtype == "bf16" else 0)
    eng = StepEngine(c, dev)
    eng.allocate()
    init_engine(eng)
    step = DataParallelStep(eng, dist, overlap_optimizer=True, bf16_wire=a.bf16_wire, comm_stream=a.comm_stream)
    batch = synthetic.make_batch(B, cfg["vocab_size"], cfg["latent_dim"], seed=shard_seed(1234, rank),
                                 image_hw=32 if cfg.get("num_regions") else 224)
    if cfg.get("num_regions"):
        batch["images"] = region_features(B, cfg["num_regions"], cfg["region_dim"], shard_seed(1234, rank))
    if len_context == 3:      # --input_mode cat (train_iq.py:72-75): the context is [<start>, category, <end>]
        batch["answers"] = batch["answer_types_for_input"]
    keys = ("images", "answers", "posteriors", "questions")
    d = {k: batch[k].to(dev) for k in keys}
    gen = torch.Generator(device=dev).manual_seed(99 + rank)

    def draw_eps():
        """Latent noise [B, engine latent width]: N(0, 1) in the real columns, zeros in the pad columns of a padded layout."""
        e_ = torch.randn(B, cfg["latent_dim"], device=dev, generator=gen)
        return e_ if eZ == cfg["latent_dim"] else torch.nn.functional.pad(e_, (0, eZ - cfg["latent_dim"]))

    # The conv stack runs one batch ahead on the step's FOURTH stream: the command processor runs at most four queues truly side by side
    # (profiles/r03_queues_exp.py: a fifth stream's kernels are time-sliced even with GPU_MAX_HW_QUEUES=8) and the step already uses the
    # caller's stream + two engine streams.  With N > 1 that fourth stream is the communication stream, which is idle until the first
    # gradient bucket is final — where the next batch's stack runs — so DataParallelStep hands it to the engine for both.
    use_prefetch = (not a.no_prefetch) and not cfg.get("num_regions")
    conv_s = step.conv_stream if (use_prefetch and dist is not None) else None
    if use_prefetch and dist is None:
        # ONE stream for everything that runs a batch ahead (the conv stack; in the PCIe-fed loop also the copy in front of it), created
        # once: a stream that is destroyed does not give its hardware queue back, and a fifth queue is time-sliced against the others
        conv_s = torch.cuda.Stream(device=dev)
        eng.adopt_conv_stream(conv_s)
    # ---- the PCIe-inclusive feed: pinned host batch, copied one step ahead on a copy stream into two device buffers -----------
    h2d_state = {}

    def h2d_setup():
        if h2d_state:
            return
        h2d_state["host"] = {k: batch[k].contiguous().pin_memory() for k in keys}
        h2d_state["bufs"] = [{k: torch.empty_like(d[k]) for k in keys} for _ in range(2)]
        # With the conv stack one batch ahead, the next batch's PCIe copy and its conv stack are ONE pipeline on the engine's conv stream
        # (copy, then the stack, both underneath the current step): no fifth stream.  Without it: a copy stream of its own, stack inline.
        h2d_state["stream"] = conv_s if use_prefetch else torch.cuda.Stream(device=dev)
        h2d_state["ready"] = [torch.cuda.Event(), torch.cuda.Event()]       # buffer filled
        h2d_state["free"] = [torch.cuda.Event(), torch.cuda.Event()]        # buffer consumed by its step
        for ev in h2d_state["free"]:
            ev.record()

    def upload(slot):
        hs = h2d_state
        with torch.cuda.stream(hs["stream"]):
            hs["stream"].wait_event(hs["free"][slot])
            for k in keys:
                hs["bufs"][slot][k].copy_(hs["host"][k], non_blocking=True)
            hs["ready"][slot].record(hs["stream"])
            if use_prefetch:
                eng.prefetch_images(hs["bufs"][slot]["images"])      # (ordered behind the copy: same stream)

    feed = {"kind": None}

    def switch_feed(kind, phase2, first):
        """The resident and the PCIe-fed loops each keep their own look-ahead pipeline: changing over consumes what the other left."""
        if feed["kind"] == kind:
            return
        while use_prefetch and eng.prefetch_pending():
            eps0 = draw_eps() if phase2 else None
            step.run(None, d["answers"], d["posteriors"], d["questions"], eps0, phase2, seed=998, kl_weight=HP["kl_weight"], lr=HP["lr"],
                     max_norm=HP["max_norm"])
        feed["kind"] = kind
        if kind == "h2d":
            h2d_setup()
            upload(first % 2)

    def one_step(i, phase2, h2d=False):
        eps = draw_eps() if phase2 else None
        switch_feed("h2d" if h2d else "resident", phase2, i)
        if h2d:
            slot = i % 2
            upload(slot ^ 1)                                   # next step's batch crosses PCIe (and runs its conv stack) underneath this step
            torch.cuda.current_stream().wait_event(h2d_state["ready"][slot])
            cur = h2d_state["bufs"][slot]
            step.run(None if use_prefetch else cur["images"], cur["answers"], cur["posteriors"], cur["questions"], eps, phase2, seed=1000 + i,
                     kl_weight=HP["kl_weight"], lr=HP["lr"], max_norm=HP["max_norm"])
            h2d_state["free"][slot].record()
            return
        # image mode: the frozen conv stack of the NEXT batch is enqueued one batch ahead (DataParallelStep.run(next_images=...)); every
        # step still contains exactly one conv stack and one of everything else
        step.run(d["images"], d["answers"], d["posteriors"], d["questions"], eps, phase2, seed=1000 + i,
                 kl_weight=HP["kl_weight"], lr=HP["lr"], max_norm=HP["max_norm"], next_images=d["images"] if use_prefetch else None)

    def timed_loop(n, first, phase2, h2d=False, profile_step=-1):
        """n steps bracketed by barrier + synchronize on both sides; returns the MAX over ranks of the wall time."""
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]      # step boundaries on the step's stream (median_ms_per_step)
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(n):
            eng.profile_enable((3 | (PROFILE_STRIDE << 8)) if i == profile_step else 0)
            one_step(first + i, phase2, h2d)
            marks[i + 1].record()
        step.finish()                                           # the last (overlapped) optimiser update is part of the timed region
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.profile_enable(0)
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n))
        timed_loop.median_ms = per_step[n // 2]
        if dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    phase2 = a.phase == 2
    print("[bench] rank %d: engine ready (workspace %.2f GB), warming up" % (rank, eng.workspace_bytes / 1e9), file=sys.stderr, flush=True)
    if a.autotune:
        # measure, don't guess: the first warm-up step times every candidate GEMM/conv kernel (tile shape, LDS-DMA ring vs
        # register staging) on the real operands of each distinct launch and caches the fastest (csrc/gemm.hip::autotune)
        eng.lib.bltvqg_debug_set(2, 1)
        one_step(0, phase2, a.h2d)
        torch.cuda.synchronize()
        eng.lib.bltvqg_debug_set(2, 0)
    for i in range(a.warmup):
        one_step(i, phase2, a.h2d)
    torch.cuda.synchronize()
    print("[bench] rank %d: warm-up done" % rank, file=sys.stderr, flush=True)
    # ---- the timed region: EXACTLY a.steps steps.  ONE of them (the middle one) carries a HIP event pair around every launch of the
    # two dominant kernel families, on the stream of the launch — an event record costs its stream a ~5.7 us bubble, ~1100 of them
    # would add ~50 % to every step, so they are sampled; the profiled step IS inside the timed region and counted in `value`.
    prof_step = a.steps // 2
    dt = timed_loop(a.steps, a.warmup, phase2, a.h2d, profile_step=prof_step)
    median_ms = timed_loop.median_ms      # GPU time between consecutive step boundaries on the step's stream (one profiled step among them)
    print("[bench] rank %d: %d timed steps in %.3f s" % (rank, a.steps, dt), file=sys.stderr, flush=True)
    conv_ms, conv_n, conv_flops = eng.profile_read(0)
    conv_by_stream = list(getattr(eng, "last_profile_by_stream_ms", []))
    gemm_ms, gemm_n, gemm_flops = eng.profile_read(1)
    gemm_by_stream = list(getattr(eng, "last_profile_by_stream_ms", []))
    stats = eng.stats()
    # What an event bracket measures on top of the kernel it brackets: empty brackets on the same stream (the two markers' own
    # latency, ~4-5 us).  It is subtracted per launch below; the raw figure is reported next to it.
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    tiny = torch.zeros(64, device=dev)
    for ea, eb in pairs:
        tiny.add_(1.0)                       # a kernel in front, as in the step (markers behind an idle stream are cheaper)
        ea.record()
        eb.record()
    torch.cuda.synchronize()
    null_us = sorted(ea.elapsed_time(eb) * 1e3 for ea, eb in pairs)[len(pairs) // 2]

    extras = {}
    if not a.no_extras and a.extra_steps > 0:
        n = a.extra_steps
        # both training phases (train_iq.py:108-111) and the PCIe-inclusive feed (train_iq.py:67-79), each as a short loop of its own
        other = not phase2
        for _ in range(2):
            one_step(0, other, a.h2d)
        dt_o = timed_loop(n, 0, other, a.h2d)
        extras["phase%d" % (2 if other else 1)] = {"ms_per_step": round(dt_o / n * 1e3, 3), "pairs_per_s": round(B * world * n / dt_o, 1), "steps": n}
        for _ in range(2):
            one_step(0, phase2, not a.h2d)
        dt_h = timed_loop(n, 0, phase2, not a.h2d)
        extras["resident" if a.h2d else "h2d"] = {
            "ms_per_step": round(dt_h / n * 1e3, 3), "pairs_per_s": round(B * world * n / dt_h, 1), "steps": n,
            "what": "same step, batch resident in HBM" if a.h2d else
                    "same step fed from a pinned host batch copied over PCIe every step, one step ahead on a copy stream (never the headline value)"}
    if use_prefetch and not a.h2d and not a.no_extras and a.extra_steps > 0:
        # How much of the conv stack the look-ahead hides: the stack alone and the rest of the step alone, each between
        # device synchronisations (median of 8), against the overlapped steady state of the timed loop.
        import statistics
        switch_feed("resident", phase2, 0)
        tc, tch = [], []
        for i in range(10):
            eps = draw_eps() if phase2 else None
            if eng.prefetch_pending() == 0:
                eng.prefetch_images(d["images"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step.run(None, d["answers"], d["posteriors"], d["questions"], eps, phase2, seed=2000 + i, kl_weight=HP["kl_weight"], lr=HP["lr"],
                     max_norm=HP["max_norm"])
            step.finish()
            torch.cuda.synchronize()
            tch.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter()
            eng.prefetch_images(d["images"])
            torch.cuda.synchronize()
            tc.append((time.perf_counter() - t0) * 1e3)
        conv_ms, chain_ms, ov_ms = statistics.median(tc[2:]), statistics.median(tch[2:]), dt / a.steps * 1e3
        extras["overlap"] = {"conv_stack_alone_ms": round(conv_ms, 3), "rest_of_step_alone_ms": round(chain_ms, 3), "overlapped_ms": round(ov_ms, 3),
                             "conv_stack_hidden_frac": round(max(0.0, conv_ms + chain_ms - ov_ms) / conv_ms, 3),
                             "what": "frozen conv stack of batch i+1 (20 convolutions + BatchNorm2d + pool) on the look-ahead stream vs everything else of "
                                     "step i; the two share the per-CU L2->LDS intake, DESIGN.md section 5c"}
    if dist and not a.no_extras and a.extra_steps > 0:
        # exposed communication per step (HIP events: end of backward on the step's stream -> end of the last all-reduce on the
        # communication stream), measured in a loop of its own so that the event pairs are not inside the headline's timed region
        step.measure_exposed = True
        for i in range(a.extra_steps):
            one_step(i, phase2, a.h2d)
        step.finish()
        torch.cuda.synchronize()
        starts = step.bucket_start_ms()
        ex = sorted(step.exposed_ms())
        step.measure_exposed = False
        if starts:
            mid = starts[len(starts) // 2]
            extras["bucket_ready_ms_before_backward_end"] = {
                "by_collective": [{"bucket": i, "ms": v} for i, v in mid],
                "what": "one measured step: how long before the END of backward each gradient bucket's collective could start (HIP events on the "
                        "communication stream behind the bucket's wait); collectives run in this order, a value near 0 = final only with backward"}
        extras["exposed_comm_ms_per_step"] = {"median": round(ex[len(ex) // 2], 3), "max": round(ex[-1], 3), "steps": len(ex),
                                              "what": "HIP events: end of backward (step stream, all engine streams joined) -> end of the last gradient "
                                                      "all-reduce (communication stream); 0 = the exchange finished under backward"}
    if world == 1 and a.f32_steps > 0 and not a.no_extras and a.dtype == "bf16":
        # reference-precision leg: the SAME step on the fp32 engine (fp32 storage, exact-fp32 MFMA) — the engine that meets the 1e-3 parity bar
        c32 = engine_config(0)
        e32 = StepEngine(c32, dev)
        e32.allocate()
        init_engine(e32)
        s32 = DataParallelStep(e32, None, overlap_optimizer=True)

        def step32(i):
            eps = draw_eps() if phase2 else None
            s32.run(d["images"], d["answers"], d["posteriors"], d["questions"], eps, phase2, seed=1000 + i, kl_weight=HP["kl_weight"],
                    lr=HP["lr"], max_norm=HP["max_norm"], next_images=d["images"] if use_prefetch else None)
        for i in range(3):
            step32(i)
        s32.finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.f32_steps):
            step32(3 + i)
        s32.finish()
        torch.cuda.synchronize()
        dt32 = time.perf_counter() - t0
        extras["f32"] = {"ms_per_step": round(dt32 / a.f32_steps * 1e3, 3), "pairs_per_s": round(B * a.f32_steps / dt32, 1), "steps": a.f32_steps,
                         "what": "same config, batch and phase on the fp32 engine (fp32 storage, exact-fp32 MFMA v_mfma_f32_16x16x4_f32): the precision that "
                                 "meets the north-star 1e-3 loss tolerance (tests/test_fullsize_gpu.py)"}
        del s32, e32
    comm = None
    if dist:
        # the exchange in isolation: each collective of the step's plan alone on the communication stream, median of 5
        comm = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "wire": "bf16" if a.bf16_wire else "fp32",
                "comm_stream": step.comm_stream_mode,
                "env": {k: v for k, v in sorted(os.environ.items())
                        if k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY") or k.startswith("NCCL_") or k.startswith("RCCL_")},
                "bucket_MB": [round(n * 4 / 2 ** 20, 2) for _, n, _ in step.buckets],
                "bucket_late": [int(late) for _, _, late in step.buckets],
                "exposed_comm_ms_per_step": extras.get("exposed_comm_ms_per_step"),
                "bucket_ready_ms_before_backward_end": extras.get("bucket_ready_ms_before_backward_end"),
                "rehearsal_note": "one-rank / shared-GPU runs move no bytes over xGMI: their step (7.5 ms on BASELINE configs[2] against 7.0 without an "
                                  "exchange) is the efficiency CEILING of this design before a byte crosses a link, not a scaling measurement",
                "allreduce": []}
        from bltvqg_amd.trainer import allreduce_bucket
        for ids, off, n in comm_plan(step.buckets, phase2):
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(step.comm):
                    e0.record()
                    allreduce_bucket(dist, eng.flat_grad, off, n, step.wire)
                    e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = sorted(ts[1:])[2]
            nbytes = n * (2 if a.bf16_wire else 4)
            w = dist.get_world_size()
            comm["allreduce"].append({"buckets": ids, "bytes": nbytes, "ms": round(ms, 3),
                                      "bus_GBps": round(2.0 * (w - 1) / w * nbytes / (ms * 1e-3) / 1e9, 1) if w > 1 else None})

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = B * world * a.steps / dt

        def family(ms_total, launches, flops, kernel, by_stream):
            if not launches:
                return None
            raw_us = ms_total * 1e3 / launches
            us = max(raw_us - null_us, 1e-3)
            # The step runs these launches on up to three streams SIDE BY SIDE, so their durations overlap in time: the contract's figure
            # (flops / sum of launch durations) is kept as `frac_sum_of_launch_durations` — it prices a launch at the whole chip although
            # it shares the CUs with the other streams' launches, and its time sum can exceed the step — while `frac` is bounded by the
            # clock: flops / max(busiest stream's bracketed time, the time sum capped at the step).  Each stream's own sum fits inside
            # the step (`per_stream_kernel_ms`, checked: `fits_in_step`).
            sum_ms = us * launches * 1e-3
            per_stream = [round(max(x - null_us * 1e-3 * (launches * x / max(ms_total, 1e-9)), 0.0), 3) for x in by_stream] if by_stream else []
            clock_ms = min(sum_ms, ms) if not per_stream else max(max(per_stream), min(sum_ms, ms))
            achieved_sum = flops / (sum_ms * 1e-3) / 1e12
            achieved = flops / (clock_ms * 1e-3) / 1e12
            return {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                    "frac_basis": "flops / min(sum of this family's bracketed launch durations, step wall time): never more time than the step has",
                    "frac_sum_of_launch_durations": round(achieved_sum / MFMA_PEAK_TFLOPS, 4),
                    "per_stream_kernel_ms": per_stream, "per_stream_order": ["caller's stream", "side 0", "side 1", "conv look-ahead"],
                    "fits_in_step": bool(all(x <= ms * 1.02 for x in per_stream)), "step_ms": round(ms, 3),
                    "traffic": None,
                    "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes)",
                    "kernel": kernel, "launches_per_step": launches, "gflop_per_step": round(flops / 1e9, 1),
                    "avg_launch_us": round(us, 2), "avg_bracket_us_raw": round(raw_us, 2), "empty_bracket_us": round(null_us, 2),
                    "profiled_steps": "1 of the %d timed steps (step %d) carries HIP event pairs on the stream of the launch: around every "
                                      "convolution and every grouped weight-gradient launch, around every %d-th plain Linear GEMM launch "
                                      "(counted %d times: a pair is a ~5 us bubble on its stream)" % (a.steps, prof_step, PROFILE_STRIDE, PROFILE_STRIDE),
                    "kernel_time_ms_per_step": round(us * launches * 1e-3, 3)}
        roof = family(gemm_ms, gemm_n, gemm_flops, by_stream=gemm_by_stream, kernel=
                      "every Linear-layer GEMM of the step (attention q|k|v / output projections, FFN, embedding, vocabulary projection, latent "
                      "nets: forward gemm_nt2_kernel<Nt2<BM,BN,..>> (gemm_dma_kernel<*,*,plain,*> for M < 256), input gradients through the transposed weight shadow, weight "
                      "gradients gemm_kernel<bf16,*,*,T,T> / wgrad_group_kernel); flops = 2*M*N*K per launch")
        roof_conv = family(conv_ms, conv_n, conv_flops, by_stream=conv_by_stream, kernel=
                           "the 20 convolution launches of the ResNet-18 stack: conv3x3_pp_kernel (13, LDS-patch 3x3), conv_stem_pool_kernel (1: stem + max-pool, pooled extrema), "
                           "gemm_dma_kernel<*,*,conv,*> (6: stride-2 / 1x1); flops over real pixels, unpadded Cin")
        # HBM bytes per launch and the rocprofv3 launch durations come from the COMMITTED profile passes of this same command (separate
        # --pmc FETCH_SIZE / WRITE_SIZE runs cannot share a process with the timed loop): static numbers, labelled with their source
        for prof_name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            pmc = os.path.join(ROOT, "profiles", prof_name)
            if os.path.exists(pmc):
                break
        if a.config == "big" and B == 256 and a.dtype == "bf16" and os.path.exists(pmc):
            tr = json.load(open(pmc))
            src = "profiles/%s (static: collected by profiles/%s at commit %s, not measured in this run)" % (
                os.path.basename(pmc), tr.get("collected_by", "collect_r02.sh"), tr.get("commit", "c1b9ef8-or-earlier"))
            if roof and "gemm_bytes_per_launch" in tr:
                roof["traffic"] = int(tr["gemm_bytes_per_launch"])
                roof["traffic_source"] = src
            if roof_conv and "conv_bytes_per_launch" in tr:
                roof_conv["traffic"] = int(tr["conv_bytes_per_launch"])
                roof_conv["traffic_source"] = src
            # the committed rocprofv3 --kernel-trace --stats run of this same command: under the profiler the host enqueue is slower,
            # the streams of the step overlap less and a launch is stretched less by the other streams' workgroups than in the
            # un-profiled brackets above
            for r_, key in ((roof, "gemm"), (roof_conv, "conv")):
                rp = (tr.get(key) or {}).get("rocprof_avg_launch_us")
                if r_ and rp:
                    r_["rocprof_avg_launch_us"] = rp
                    r_["rocprof_frac"] = round(r_["gflop_per_step"] / r_["launches_per_step"] / rp * 1e3 / r_["peak"], 4)      # GFLOP / us = PFLOP/s
                    r_["rocprof_source"] = src
        out = {
            "metric": "image-question pairs/sec (train step)", "value": round(value, 1), "unit": "pairs/s", "n_gpus": world,
            "world_size": (dist.get_world_size() if dist is not None else 1),
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "median_ms_per_step": round(median_ms, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype,
            "data": ("synthetic, pinned host batch copied over PCIe every step (one step ahead)" if a.h2d else "synthetic") +
                    (" — REHEARSAL: %d ranks share %d GPU(s) over %s, not a measurement" % (world, torch.cuda.device_count(), backend)
                     if (share_gpu or backend != "nccl") else ""),
            "config": {"workload": "%s: IQ train step (fwd+loss+bwd+clip+Adam), %s cfg: %d-layer d_model=%d %d-head, per-GPU batch %d, %s, "
                                   "T=20/S_a=5/S_p=21, V=%d, phase %d, dropout 0.1" % (
                                       CONFIG_NAMES[a.config], a.config, cfg["num_layers"], cfg["hidden_dim"], cfg["num_heads"], B,
                                       "%dx%d region features (no CNN)" % (cfg["num_regions"], cfg["region_dim"]) if cfg.get("num_regions")
                                       else "224x224 images", cfg["vocab_size"], a.phase),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "loss_rec": round(stats["rec"], 4),
                       "model_tflops": round(value * FLOP_PER_PAIR[a.config] / 1e12, 2)},
            "roofline": roof if roof else roof_conv,
        }
        if roof and roof_conv:
            out["roofline_conv"] = roof_conv
        out["debug_keys"] = debug_keys
        out["conv_stack"] = ("one batch ahead on the engine's conv stream (bltvqg_engine_prefetch_images)" if use_prefetch and not a.h2d
                             else "inline in forward")
        out.update(extras)
        if comm:
            out["dist"] = comm
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, phase2, a.cpu_batch, a.cpu_steps)
            out["loss_vs_oracle"] = loss_vs_oracle(eng, cfg, a, B, phase2, batch, rank, engine_config) if pad is None else {
                "skipped": "padded engine layout (parameters live at padded positions): this configuration is held to the oracle by "
                           "tests/test_fullsize_gpu.py::test_reference_cli_defaults_at_their_own_depth_and_batch"}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
