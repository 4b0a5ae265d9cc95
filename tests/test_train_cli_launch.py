"""`python train_iq.py --synthetic --num_gpus N` is how a user of the reference gets data parallelism (train_iq.py:349,372-373:
pl.Trainer(gpus=args.num_gpus)).  CPU rehearsal of that entry point: the parent spawns N ranks BEFORE any GPU call, every rank joins the
process group (gloo here, RCCL on the GPUs), draws its own shard of every global batch and trains through the real DataParallelStep;
the replicas end bit-identical."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, n):
    env = dict(os.environ, BLT_DIST_BACKEND="gloo", BLT_TRAINER_FACTORY="dp_stub:factory", BLT_STUB_OUT=str(tmp_path / "out"),
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, os.environ.get("PYTHONPATH", "")]))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "train_iq.py"), "--synthetic", "--use_gpu", "", "--num_gpus", str(n), "--batch_size", "4",
           "--total_training_steps", "5", "--num_pretraining_steps", "2", "--hidden_dim", "64", "--latent_dim", "64", "--pwffn_dim", "128"]
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def test_num_gpus_flag_spawns_ranks_that_train_to_identical_replicas(tmp_path):
    out = _run(tmp_path, 2)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "spawning 2 ranks" in out.stderr
    r0 = torch.load(str(tmp_path / "out.rank0.pt"), weights_only=True)
    r1 = torch.load(str(tmp_path / "out.rank1.pt"), weights_only=True)
    assert r0["world"] == 2 and r1["world"] == 2
    assert torch.equal(r0["params"], r1["params"])                      # replicas bit-identical after 5 steps across the phase switch
    assert float(r0["params"].abs().max()) > 0.0                        # ... and they did train
    assert r0["batches"] != r1["batches"] and len(r0["batches"]) == 5   # every rank drew its own shard of each global batch


def test_single_process_runs_no_launcher(tmp_path):
    out = _run(tmp_path, 1)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "spawning" not in out.stderr
    r0 = torch.load(str(tmp_path / "out.rank0.pt"), weights_only=True)
    assert r0["world"] == 1 and len(r0["batches"]) == 5


def test_parent_spawns_before_anything_touches_the_gpu():
    src = open(os.path.join(ROOT, "blt-vqg_amd", "train_iq.py")).read()
    body = src[src.index("def main(argv=None):"):]
    assert body.index("spawn_ranks(") < body.index("torch.cuda.is_available()")
    launch = open(os.path.join(ROOT, "blt-vqg_amd", "launch.py")).read()
    assert "import torch" not in launch and "_lib" not in launch
