"""bench.py's one-JSON-line contract (the driver parses it): helper logic on CPU, a real tiny run on the GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def test_roofline_helper_and_traffic_lookup():
    sys.path.insert(0, ROOT)
    import bench
    tm = {"fit_ms": 70.0, "fits": 2,
          "ms": {"xty": 1.4, "xb": 0.0, "deflate": 0.0, "fused": 66.0, "small": 0.8},
          "launches": {"xty": 2, "xb": 0, "deflate": 0, "fused": 40, "small": 80},
          "bytes": {"xty": 2 * 4303360000, "xb": 0, "deflate": 0, "fused": 40 * 8391556096, "small": 0}}
    r = bench.roofline_of(tm)
    assert r["kernel"] == "fused" and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - 8391556096 / (66.0 / 40 * 1e-3) / 1e9) < 0.1
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4 and r["traffic"] is None
    tr, src = bench.pmc_traffic("C3", "nipals", 1, "fused")
    assert src and src.startswith("profiles/") and 8.0e9 < tr < 9.0e9       # committed PMC summary of the headline plan
    assert bench.pmc_traffic("C3", "nipals", 0, "deflate") == (None, None)  # no summary for that plan
    assert bench.roofline_of({"ms": {k: 0 for k in tm["ms"]}, "launches": {k: 0 for k in tm["ms"]},
                              "bytes": {k: 0 for k in tm["ms"]}, "fits": 0, "fit_ms": 0}) is None


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1",
                        "--cpu-rows", "4096"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == "NIPALS components/sec" and d["unit"] == "components/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1


@pytest.mark.gpu
def test_bench_profile_after_times_the_fits_without_event_brackets():
    """--profile-after: the timed steps carry no HIP-event brackets (as every N > 1 run), the roofline comes from extra
    profiled steps; the line says where it was measured."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--workload", "tiny",
                        "--no-cpu", "--no-alt", "--profile-after"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert d["value"] > 0 and d["roofline"]["measured_over"].startswith("separate profiled steps")
    assert d["roofline"]["avg_launch_ms"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_as_the_driver_launches_it():
    """The driver's N > 1 launch line (torch.distributed.run, one process per rank) with the gloo rehearsal backend
    and both ranks on the one GPU of the test box: rank 0 prints exactly one line with n_gpus = 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "tiny", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["parallelism"] == "row-shard x2"
    assert d["roofline"]["measured_over"].startswith("separate profiled steps")
    # the RCCL leg: a figure with the rank count RCCL reports, or the reason there is none (two ranks on ONE GPU here)
    rl = d["alt"]["rccl"]
    assert ("components_per_s" in rl and rl["nranks_reported_by_rccl"] == 2) or "one rank per device" in rl["unavailable"], rl


@pytest.mark.gpu
def test_bench_gpus_flag_alone_starts_the_ranks():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) must not fall back to a silent
    single-GPU run: it starts the two ranks itself (child torch.distributed.run, before any GPU call) and the one
    JSON line says n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "tiny", "--backend", "gloo"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "row-shard x2" and d["config"]["reducer"] == "torch"


@pytest.mark.gpu
@pytest.mark.parametrize("algo,tail", [("nipals", None), ("kernel", None), ("nipals", "2")],
                         ids=["nipals", "kernel-plan", "nipals-update-in-the-tail"])
def test_bench_two_ranks_over_the_device_side_exchange(algo, tail):
    """The reducer `bench.py --gpus N` takes by default on a multi-GPU node (`--reducer ipc`: the library's device-side
    exchange between processes), rehearsed with two ranks on the one GPU (gloo only carries the set-up): one contract
    line, the reducer it reports is the exchange itself (no fall-back), and the replica guard at the end stayed quiet.
    Under the NIPALS plan (the headline), the KERNEL plan (what a `PLS::Model` takes by default) and with the one-response
    update run inside the pass's tail (PLS_HIP_TAIL=2: one launch per sharded component)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if tail:
        env["PLS_HIP_TAIL"] = tail
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "tiny", "--backend", "gloo", "--reducer", "ipc", "--algo", algo, "--no-alt"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["reducer"] == "ipc" and d["value"] > 0


@pytest.mark.gpu
def test_bench_rccl_leg_code_path_at_one_rank():
    """The RCCL leg of an N > 1 line (alt.rccl) can not run with two ranks on one GPU; its code path -- nccl process group,
    the library's own communicator attached behind the headline, ncclCommCount, three timed fits -- is exercised here with ONE
    rank: the leg must report RCCL's rank count and a rate."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "tiny",
                        "--no-cpu", "--rccl-leg-at-one-rank"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rl = json.loads(lines[0])["alt"]["rccl"]
    assert rl.get("nranks_reported_by_rccl") == 1 and rl["components_per_s"] > 0, rl


@pytest.mark.gpu
def test_bench_falls_back_when_the_exchange_fails_in_the_fits():
    """A reducer that passed its set-up can still fail in the fits (the device-side exchange has never crossed a real xGMI
    link in this pipeline).  Fault injection: rank 1 never delivers its first partial sums, every rank's wait ends at the
    time limit and the fits come back poisoned -- the ranks must agree on that, drop the exchange, take the torch reducer,
    measure again and still print the one line, which names what happened."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", PLS_HIP_XCHG_TIMEOUT_S="1.5", PLS_HIP_TEST_DROP_PUSH="1:2",
               PLS_AMD_LIBRARY=os.path.join(ROOT, "pls_amd", "csrc", "testing", "libpls_hip.so"))  # (collective 1 is the self-test)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "tiny", "--backend", "gloo", "--reducer", "ipc", "--no-alt"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["reducer"].startswith("torch (the ipc reducer failed in the fits"), d["config"]["reducer"]
