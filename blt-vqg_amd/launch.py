"""One process per GPU, started from ONE command (the reference's `pl.Trainer(gpus=args.num_gpus)`, train_iq.py:349,372-373).

`spawn_ranks(n, script, argv)` starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 ... script argv`
as a CHILD process and relays its exit code.  Only the standard library is imported here: the parent must not have initialised the GPU
(no torch.cuda call, no HIP library load) — replacing or forking a GPU-initialised process is not allowed on the GPU boxes, and the
ranks inherit nothing from the parent but the environment.
"""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher():
    """True inside a rank started by torch.distributed.run (or any launcher that exports WORLD_SIZE / RANK)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def rank_env():
    """(rank, world, local_rank) of this process; (0, 1, 0) outside a launcher."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def spawn_command(n, script, argv, port=None):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(script)] + list(argv)


def spawn_ranks(n, script, argv, relay_prefix=None, env=None):
    """Runs the ranks to completion.  relay_prefix: only stdout lines starting with it go to this process's stdout (the last one; everything
    else to stderr) — bench.py's one JSON line; None: the children's stdout is inherited.  Returns the exit code."""
    cmd = spawn_command(n, script, argv)
    print("[launch] spawning %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL / tensor sharing across processes)
    if relay_prefix is None:
        return subprocess.call(cmd, env=e)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=e)
    line = None
    for ln in p.stdout:
        if ln.startswith(relay_prefix):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line:
        print(line, flush=True)
    return rc if rc else (0 if line else 1)
