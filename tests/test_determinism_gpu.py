"""Race screen: the same training step repeated must reproduce its outputs exactly and its losses / gradients / parameters up to
float-atomic accumulation order.  (Caught a real bug: the CE kernel read the target logit after the barrier that precedes the
in-place gradient write; only the loss statistic was affected, intermittently.)"""
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [0, 1])
def test_repeated_step_is_reproducible(dtype):
    from bltvqg_amd.engine import StepEngine, make_config
    z, cfg, state, batch = load_golden("tiny")
    c = make_config(4, cfg.hidden_dim, cfg.pwffn_dim, cfg.latent_dim, cfg.emb_dim, cfg.num_layers, cfg.num_heads, cfg.vocab_size,
                    image_hw=(64, 64), dtype=dtype, attention_dropout=0.1, relu_dropout=0.1)
    e = StepEngine(c)
    e.allocate()
    d = {k: v.cuda() for k, v in batch.items()}
    ref = None
    for rep in range(60):
        e.load_state(state)
        e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"], True, 7)
        out = e.read(0)
        e.loss_backward(0.3)
        st = e.read(4).clone()
        g = e.flat_grad.clone()
        e.optimizer_step(1e-4, 5.0)
        p = e.flat_train.clone()
        torch.cuda.synchronize()
        cur = (out.cpu(), st.cpu(), g.cpu(), p.cpu())
        if ref is None:
            ref = cur
            continue
        assert torch.equal(cur[0], ref[0]), rep                                   # forward is bit-reproducible
        assert float((cur[1][:4] - ref[1][:4]).abs().max()) < 1e-4, (rep, (cur[1] - ref[1]).tolist())
        assert float((cur[2] - ref[2]).abs().max()) <= 1e-4 * float(ref[2].abs().max()), rep
        assert float((cur[3] - ref[3]).abs().max()) < 1e-5, rep


def test_forward_is_bit_reproducible_with_folded_layernorms_at_d_model_512():
    """The folded LayerNorms take their row sums from the producing GEMM's epilogue, one partial per column tile (8 of them at
    d_model 512), added by the consumer in slot order: no float atomics, so the forward stays bit-reproducible at the benchmark widths
    (the fixture above has one column tile per row)."""
    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import init_reference_style
    B = 16
    c = make_config(B, 512, 2048, 512, 300, 2, 8, 8000, image_hw=(64, 64), dtype=1, attention_dropout=0.1, relu_dropout=0.1)
    e = StepEngine(c)
    e.allocate()
    init_reference_style(e, seed=1)
    b = synthetic.make_batch(B, 8000, 512, seed=5, image_hw=64)
    d = {k: v.cuda() for k, v in b.items() if torch.is_tensor(v)}
    ref = None
    for rep in range(25):
        e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"], True, 7)
        out = (e.read(0).clone(), e.read(1).clone(), e.read(3).clone())
        torch.cuda.synchronize()
        if ref is None:
            ref = out
            continue
        for a, r in zip(out, ref):
            assert torch.equal(a, r), rep
