#!/usr/bin/env python3
"""Headline benchmark: image-question pairs/s of the BLT-VQG IQ train step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one full reference training step (train_iq.py:105-132 + Lightning's backward / clip 5 / Adam) on one synthetic
minibatch per GPU, inputs already resident in HBM: frozen ResNet-18 forward (train-mode BatchNorm), both transformer
encoders, latent, decoder, losses, the whole backward, gradient all-reduce (N > 1), clip + Adam.  Dropout is on at the
reference's 0.1/0.1.  Workload at N=1 = BASELINE.json configs[1]: the 2-layer d_model=256 model at batch 128, bf16.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP multiplexes streams onto 4 hardware queues by default; the step uses the main stream, two engine side streams and (N > 1) the
# trainer's communication stream plus RCCL's own: with 8 queues none of them is silently serialised behind another
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[0]/[1]: train_iq.py defaults-shaped model (small)
    "small": dict(hidden_dim=256, pwffn_dim=512, latent_dim=256, emb_dim=300, num_layers=2, num_heads=4, vocab_size=8000, batch=128),
    # BASELINE.json configs[2]/[3]: 6-layer d_model=512 8-head (big)
    "big": dict(hidden_dim=512, pwffn_dim=2048, latent_dim=512, emb_dim=300, num_layers=6, num_heads=8, vocab_size=8000, batch=256),
    # BASELINE.json configs[4]: bottom-up features (36 x 2048 regions, no CNN), 6-layer transformer, global batch 512 = 8 x 64
    "regions": dict(hidden_dim=512, pwffn_dim=2048, latent_dim=512, emb_dim=300, num_layers=6, num_heads=8, vocab_size=8000, batch=64,
                    num_regions=36, region_dim=2048),
}
# algorithmic FLOP per pair per train step: CNN fwd x1 + everything trainable x3; regions: the projection runs on the
# region mean (the mean commutes with the Linear), 2.1 MFLOP per pair instead of the survey's 75.5
FLOP_PER_PAIR = {"small": 4.25e9, "big": 9.95e9, "regions": 6.33e9}


def region_features(B, R, D, seed):
    """Synthetic bottom-up features: non-negative (post-ReLU) with a per-sample component."""
    g = torch.Generator().manual_seed(int(seed) + 77)
    x = torch.relu(torch.randn(B, R, D, generator=g) + 0.3 * torch.randn(B, 1, D, generator=g))
    return (x * (0.5 + torch.rand(B, 1, 1, generator=g))).contiguous()
PROFILE_EVERY = 10
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="small", choices=sorted(CONFIGS))
    ap.add_argument("--h2d", action="store_true",
                    help="PCIe-inclusive variant (NOT the headline value): the batch lives in pinned host memory and is copied to the GPU "
                         "every step on a copy stream, one step ahead (the reference's boundary hands over host tensors, train_iq.py:67-79)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--phase", type=int, default=2, choices=[1, 2], help="1 = pre-training (latent off), 2 = latent on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--autotune", action="store_true", help="time candidate GEMM/conv kernels per launch in the first warm-up step")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=3)
    return ap.parse_args()


def cpu_baseline(cfg, phase2, batch, steps):
    """The CPU oracle (plain PyTorch fp32 restatement of the reference, pinned by tests/golden) timed on the host cores:
    full train step (forward, losses, backward, clip, Adam) on the same synthetic workload at a bounded batch."""
    from types import SimpleNamespace
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import iq_oracle as O
    from synth import synth_state
    import bltvqg_amd.synthetic as synthetic
    ns = SimpleNamespace(emb_dim=cfg["emb_dim"], hidden_dim=cfg["hidden_dim"], latent_dim=cfg["latent_dim"], pwffn_dim=cfg["pwffn_dim"],
                         num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], vocab_size=cfg["vocab_size"],
                         num_regions=cfg.get("num_regions", 0), region_dim=cfg.get("region_dim", 0))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))         # the GPU box gives a 1-GPU job a 16-core share; more threads only oversubscribe
    torch.set_num_threads(cores)
    print("[bench] cpu baseline: %d threads (os.cpu_count() = %s)" % (cores, os.cpu_count()), file=sys.stderr, flush=True)
    state = synth_state(O.iq_spec(ns), seed=1)
    P = O.clone_params(state)
    names = O.trainable_names(P)
    opt = torch.optim.Adam([P[n] for n in names], lr=1e-4)
    hp = O.default_hp()
    b = synthetic.make_batch(batch, ns.vocab_size, ns.latent_dim, seed=1234, image_hw=32 if ns.num_regions else 224)
    if ns.num_regions:
        b["images"] = region_features(batch, ns.num_regions, ns.region_dim, 1234)
    gen = torch.Generator().manual_seed(7)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        masks = None
        out, z_logit, kld, recon, _ = O.iq_forward(P, ns, b["images"], b["answers"], b["posteriors"], b["questions"], phase2,
                                                   torch.randn(batch, ns.latent_dim, generator=gen), masks, 0.0, True, {})
        loss, _ = O.calculate_losses(out, recon, kld, z_logit, b["questions"], phase2, 100, hp)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([P[n] for n in names if P[n].grad is not None], 5.0)
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
        print("[bench] cpu baseline step %d: %.2f s" % (i, time.perf_counter() - t0), file=sys.stderr, flush=True)
    t = sorted(times)[len(times) // 2]
    return dict(value=round(batch / t, 2), unit="pairs/s", cores=cores, kind="port",
                sample="%d timed train steps (median) of the CPU oracle at batch %d, fp32, dropout off, same model config and synthetic inputs"
                       % (steps, batch))


def main():
    a = parse()
    # stdout carries exactly ONE line, the JSON: native libraries that write to file descriptor 1 (RCCL prints a version banner there
    # when NCCL_DEBUG=VERSION is set, as it is on the GPU boxes) are sent to stderr for the duration of the run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("BLT_FORCE_DIST") == "1":      # BLT_FORCE_DIST: rehearse the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import DataParallelStep, init_reference_style

    # A/B switches of the kernels (bltvqg_debug_set): BLT_DEBUG="key=value,key=value"
    if os.environ.get("BLT_DEBUG"):
        from bltvqg_amd import _lib as _l
        for kv in os.environ["BLT_DEBUG"].split(","):
            k, v = kv.split("=")
            _l.load().bltvqg_debug_set(int(k), int(v))
    cfg = dict(CONFIGS[a.config])
    B = a.batch or cfg.pop("batch")
    cfg.pop("batch", None)
    phase2 = a.phase == 2
    c = make_config(B, cfg["hidden_dim"], cfg["pwffn_dim"], cfg["latent_dim"], cfg["emb_dim"], cfg["num_layers"], cfg["num_heads"],
                    cfg["vocab_size"], dtype=1 if a.dtype == "bf16" else 0, num_regions=cfg.get("num_regions", 0),
                    region_dim=cfg.get("region_dim", 0))
    eng = StepEngine(c, dev)
    eng.allocate()
    init_reference_style(eng, seed=0)                      # same weights on every rank
    step = DataParallelStep(eng, dist, overlap_optimizer=True)
    from bltvqg_amd.trainer import shard_seed
    batch = synthetic.make_batch(B, cfg["vocab_size"], cfg["latent_dim"], seed=shard_seed(1234, rank),
                                 image_hw=32 if cfg.get("num_regions") else 224)
    if cfg.get("num_regions"):
        batch["images"] = region_features(B, cfg["num_regions"], cfg["region_dim"], shard_seed(1234, rank))
    d = {k: v.to(dev) for k, v in batch.items() if k in ("images", "answers", "posteriors", "questions")}
    gen = torch.Generator(device=dev).manual_seed(99 + rank)

    keys = ("images", "answers", "posteriors", "questions")
    if a.h2d:
        host = {k: batch[k].contiguous().pin_memory() for k in keys}
        bufs = [{k: torch.empty_like(d[k]) for k in keys} for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=dev)
        ready = [torch.cuda.Event(), torch.cuda.Event()]       # buffer filled
        free = [torch.cuda.Event(), torch.cuda.Event()]        # buffer consumed by its step

        def upload(slot):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(free[slot])
                for k in keys:
                    bufs[slot][k].copy_(host[k], non_blocking=True)
                ready[slot].record(copy_stream)

        for ev in free:
            ev.record()
        upload(0)

    def one_step(i):
        eps = torch.randn(B, cfg["latent_dim"], device=dev, generator=gen) if phase2 else None
        cur = d
        if a.h2d:
            slot = i % 2
            upload(slot ^ 1)                                   # next step's batch crosses PCIe underneath this step
            torch.cuda.current_stream().wait_event(ready[slot])
            cur = bufs[slot]
        step.run(cur["images"], cur["answers"], cur["posteriors"], cur["questions"], eps, phase2, seed=1000 + i,
                 kl_weight=0.5, lr=1e-4, max_norm=5.0)
        if a.h2d:
            free[i % 2].record()

    print("[bench] rank %d: engine ready (workspace %.2f GB), warming up" % (rank, eng.workspace_bytes / 1e9), file=sys.stderr, flush=True)
    if a.autotune:
        # measure, don't guess: the first warm-up step times every candidate GEMM/conv kernel (tile shape, LDS-DMA ring vs
        # register staging) on the real operands of each distinct launch and caches the fastest (csrc/gemm.hip::autotune)
        eng.lib.bltvqg_debug_set(2, 1)
        one_step(0)
        torch.cuda.synchronize()
        eng.lib.bltvqg_debug_set(2, 0)
    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize()
    print("[bench] rank %d: warm-up done" % rank, file=sys.stderr, flush=True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    # The conv kernels are bracketed by HIP events on their own stream for the roofline figure.  An event record costs the stream
    # a ~5.7 us bubble (40 of them per step = 5 % of a 4 ms step), so only every PROFILE_EVERY-th timed step carries them.
    profiled = 0
    t0 = time.perf_counter()
    for i in range(a.steps):
        on = (i % PROFILE_EVERY == 0)
        eng.profile_enable(on)
        profiled += int(on)
        one_step(a.warmup + i)
    step.finish()                                           # the last (overlapped) optimiser update is part of the timed region
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("[bench] rank %d: %d timed steps in %.3f s" % (rank, a.steps, dt), file=sys.stderr, flush=True)
    conv_ms, conv_launches, conv_flops = eng.profile_read()
    eng.profile_enable(False)
    # What an event bracket measures on top of the kernel it brackets: empty brackets on the same stream (the two markers' own
    # latency, ~4-5 us).  It is subtracted per launch below; the raw figure is reported next to it.
    null_us = 0.0
    if conv_launches:
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        tiny = torch.zeros(64, device=dev)
        for ea, eb in pairs:
            tiny.add_(1.0)                       # a kernel in front, as in the step (markers behind an idle stream are cheaper)
            ea.record()
            eb.record()
        torch.cuda.synchronize()
        null_us = sorted(ea.elapsed_time(eb) * 1e3 for ea, eb in pairs)[len(pairs) // 2]
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = eng.stats()
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = B * world * a.steps / dt
        raw_us = conv_ms * 1e3 / max(conv_launches, 1)
        launch_us = max(raw_us - null_us, 1e-3)
        achieved = conv_flops / (launch_us * 1e-6 * max(conv_launches, 1)) / 1e12 if conv_ms > 0 else 0.0
        # HBM bytes per conv launch from the committed PMC passes (profiles/summarize_pmc.py); they were collected for this
        # default workload only, so any other configuration reports null
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_conv_traffic.json")
        if a.config == "small" and B == 128 and a.dtype == "bf16" and os.path.exists(pmc):
            traffic = int(json.load(open(pmc))["traffic_MB_per_launch"] * 1e6)
        out = {
            "metric": "image-question pairs/sec (train step)", "value": round(value, 1), "unit": "pairs/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic, pinned host batch copied over PCIe every step (one step ahead)" if a.h2d else "synthetic",
            "config": {"workload": "IQ train step (fwd+loss+bwd+clip+Adam), %s cfg: %d-layer d_model=%d, per-GPU batch %d, %s, "
                                   "T=20/S_a=5/S_p=21, V=%d, phase %d, dropout 0.1" % (
                                       a.config, cfg["num_layers"], cfg["hidden_dim"], B,
                                       "%dx%d region features (no CNN)" % (cfg["num_regions"], cfg["region_dim"]) if cfg.get("num_regions")
                                       else "224x224 images", cfg["vocab_size"], a.phase),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "loss_rec": round(stats["rec"], 4),
                       "model_tflops": round(value * FLOP_PER_PAIR[a.config] / 1e12, 2)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes)",
                         "kernel": "the 20 convolution launches of the ResNet-18 stack: conv3x3_pp_kernel (13, LDS-patch 3x3), "
                                   "conv_stem_direct_kernel (1), gemm_dma_kernel<*,*,conv,*> (6: stride-2 / 1x1)" if conv_launches else
                                   "not measured: this configuration has no convolution stack (the bracketed kernels)",
                         "launches_per_step": conv_launches // max(profiled, 1),
                         "avg_launch_us": round(launch_us, 2), "avg_bracket_us_raw": round(raw_us, 2),
                         "empty_bracket_us": round(null_us, 2),
                         "profiled_steps": "%d of the %d timed steps (every %dth) carry the HIP events" % (profiled, a.steps, PROFILE_EVERY),
                         "kernel_share_of_step": round(launch_us * 1e-3 * (conv_launches // max(profiled, 1)) / (dt / a.steps * 1e3), 4)},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, phase2, a.cpu_batch, a.cpu_steps)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
