"""bench.py --gpus N started WITHOUT a launcher must start N ranks itself (the reference gets data parallelism from one flag,
pl.Trainer(gpus=N), train_iq.py:372-373).  CPU rehearsal of that launch path: parent -> torch.distributed.run -> N ranks over gloo
-> rank 0's JSON line relayed on the parent's stdout."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_that_many_ranks():
    env = dict(os.environ, BLT_BENCH_REHEARSE_LAUNCH="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout          # stdout carries exactly one line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2


def test_parent_of_a_spawn_never_imports_the_hip_library():
    """The parent process must not initialise the GPU before it starts the ranks: the spawn branch runs before torch / the HIP
    library are imported."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    assert "import torch" not in head.replace("    import torch", "")      # only function-local imports above main()
    body = src[src.index("def main():"):]
    assert body.index("spawn_ranks(a)") < body.index("import torch")


def test_bench_refuses_work_skipping_debug_keys():
    """bench.py echoes every non-zero debug key and exits non-zero when a work-skipping ablation (keys 14 / 15) is set or the ablation
    build is loaded: a timed region that can skip work is not a measurement."""
    import importlib.util
    import os
    import pytest
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class FakeLib(object):
        def __init__(self, keys, abl=0):
            self.keys, self.abl = keys, abl

        def bltvqg_debug_get(self, k):
            return self.keys.get(k, 0)

        def bltvqg_build_has_ablations(self):
            return self.abl
    assert bench.check_debug_keys(FakeLib({})) == {}
    assert bench.check_debug_keys(FakeLib({11: 1, 3: 2})) == {"3": 2, "11": 1}
    for lib in (FakeLib({14: 1}), FakeLib({15: 1}), FakeLib({}, abl=1)):
        with pytest.raises(SystemExit) as ei:
            bench.check_debug_keys(lib)
        assert ei.value.code != 0


def test_shipped_library_has_no_work_skipping_switch():
    from bltvqg_amd import _lib
    lib = _lib.load()
    assert lib.bltvqg_build_has_ablations() == 0
    lib.bltvqg_debug_set(7, 3)
    assert lib.bltvqg_debug_get(7) == 3
    lib.bltvqg_debug_set(7, 0)


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r04_bench_default.json is the bench line of a default `python bench.py` run on an MI355X box: the driver's contract fields, the
    roofline block of the dominant kernel family and the CPU baseline must all be there (a schema check, not a performance check)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r04_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "median_ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "debug_keys", "loss_vs_oracle", "overlap", "f32", "phase1"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert "configs[2]" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_sum_of_launch_durations", "per_stream_kernel_ms", "fits_in_step", "step_ms"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] < 1.0
    # the fraction the contract reads is bounded by the step's clock — no stream's bracketed time exceeds the step,
    # and `frac` uses no more time than the step has (the plain flops / sum-of-durations figure stays beside it)
    assert r["fits_in_step"] is True and all(x <= r["step_ms"] * 1.02 for x in r["per_stream_kernel_ms"])
    assert r["frac"] >= r["frac_sum_of_launch_durations"] - 1e-9
    assert r["gflop_per_step"] * 1e9 / (r["frac"] * r["peak"] * 1e12) <= r["step_ms"] * 1e-3 * 1.001
    assert "static" in r["traffic_source"] and "r04_pmc_traffic.json" in r["traffic_source"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "threads", "affinity_cpus", "os_cpu_count", "threads_tried_s_per_step", "kind", "sample", "s_per_step"):
        assert k in c, k
    assert c["kind"] == "port" and d["debug_keys"] == {} and c["cores"] == c["threads"] <= min(c["affinity_cpus"], 64)
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) / d["value"] < 1e-3
