"""End-to-end sanity on the GPU: the fused train step (forward + losses + backward + clip + Adam, dropout on, overlapped optimiser) memorises
one fixed synthetic batch — every statistic stays finite and the reconstruction loss falls by an order of magnitude — and a forward +
backward repeated with the same seed reproduces its statistics (what `TrainIQ.fused_training_step` / bench.py drive, reference
train_iq.py:105-132)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfgname", ["small", "big_b64"])
def test_fused_step_memorises_a_fixed_batch(cfgname):
    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import DataParallelStep, init_reference_style
    C = dict(small=(128, 256, 512, 256, 300, 2, 4, 8000), big_b64=(64, 512, 2048, 512, 300, 6, 8, 8000))[cfgname]
    dev = torch.device("cuda", 0)
    eng = StepEngine(make_config(*C, dtype=1), dev)
    eng.allocate()
    init_reference_style(eng, seed=0)
    step = DataParallelStep(eng, None, overlap_optimizer=True)
    b = synthetic.make_batch(C[0], C[7], C[3], seed=4321)
    d = {k: b[k].to(dev) for k in ("images", "answers", "posteriors", "questions")}
    eps = torch.randn(C[0], C[3], device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    first = last = None
    for i in range(160):
        step.run(d["images"], d["answers"], d["posteriors"], d["questions"], eps, True, seed=i, kl_weight=0.1, lr=3e-4)
        if i % 40 == 0 or i == 159:
            step.finish()
            torch.cuda.synchronize()
            st = eng.stats()
            assert all(v == v and abs(v) < 1e6 for v in (st["rec"], st["kld"], st["img"], st["aux"], st["grad_norm"])), (i, st)
            first = first or st
            last = st
    assert last["rec"] < 0.1 * first["rec"], (first, last)
    assert last["img"] < first["img"]
    # same seed, no update in between: the statistics repeat (gradients up to the order of a few fp32 atomics)
    outs = []
    for _ in range(2):
        eng.forward(d["images"], d["answers"], d["posteriors"], d["questions"], eps, True, 77)
        eng.loss_backward(0.1)
        torch.cuda.synchronize()
        outs.append((eng.stats(), eng.flat_grad.clone()))
    assert abs(outs[0][0]["rec"] - outs[1][0]["rec"]) <= 1e-6 * max(1.0, abs(outs[0][0]["rec"]))
    gmax = float(outs[0][1].abs().max())
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-3 * gmax
