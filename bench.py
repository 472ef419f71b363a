#!/usr/bin/env python3
"""Headline benchmark: NeuMF full-catalog scoring + top-k on N MI355X (BASELINE.json configs[1] and configs[4]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for every N: typed plainly with N > 1, this process starts the N ranks itself (a child
`python -m torch.distributed.run`, before anything here touches the GPU) and relays rank 0's line.

Two workloads, both at every N (user-sharded, tables replicated, weak scaling: per-GPU work is fixed):
  * `value` -- cfg2: every rank scores its 6,038 users against the 3,533-item catalog (21,332,254 pairs: layer-1
    projections + fused tower + top-10 select) per step, then all ranks all-gather the top-10 lists over RCCL (one packed
    message; the gather of step i overlaps the kernels of step i + 1, and the last gather is inside the timed region).
    Weights: NeuMF-end random init (reference constructor order, torch.manual_seed(0)); at N > 1 the user table has
    N x 6,038 rows.  K steps after W warm-up steps, barrier + synchronize on both sides, max over ranks.
  * `cfg5_top100` -- the second half of the metric: synthetic NeuMF 10 M users x 1 M items, full-catalog top-100 users/s;
    every rank scores a 1,024-user slice per step (1.024e9 pairs) and the top-100 lists are all-gathered.  Same timing
    protocol, its own step count (--cfg5-steps).  Runs after the cfg2 region.
Inputs are resident in HBM before the timed regions.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_USERS, N_ITEMS, FACTOR, LAYERS, K = 6038, 3533, 32, 3, 10
FLOP_PER_PAIR = 2 * (256 * 128 + 128 * 64 + 64 * 32) + 32 + 2 * 64          # 86,176 algorithmic
FLOP_PER_PAIR_EXEC = 2 * (128 * 64 + 64 * 32) + 32 + 2 * 64 + 128          # 20,768 executed by the tile kernel: MFMA layers 2-3,
#                                                                              GMF, predict, and the layer-1 add (factorised layer 1)
PEAK_F32_MFMA_TFLOPS = 157.3                                                # MI355X_MICROARCH.md chip-level parameters


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the synthetic 10M x 1M top-100 slice (field cfg5_top100)")
    ap.add_argument("--cfg5", action="store_true", help=argparse.SUPPRESS)      # accepted for older command lines: now the default
    ap.add_argument("--cfg5-steps", type=int, default=3)
    ap.add_argument("--cfg5-users-per-gpu", type=int, default=1024)
    ap.add_argument("--no-split", action="store_true", help="skip the opt-in bf16x3 mode's figures (field split_bf16x3)")
    ap.add_argument("--share-gpus", action="store_true",
                    help="REHEARSAL ONLY: run the N ranks on however many GPUs are visible (ranks share devices, gloo backend "
                         "because RCCL refuses two ranks on one device); checks the N > 1 code path on a 1-GPU box, the numbers "
                         "mean nothing")
    ap.add_argument("--launch-timeout", type=float, default=600.0, help="seconds before the self-started ranks are taken to be hung")
    ap.add_argument("--precondition", type=int, default=48,
                    help="untimed steps BEFORE the W warm-up steps of each cfg2 region: the device clock needs ~30 ms of load after "
                         "idle to settle (profiles/r03_torchrun_gap.md), which W = 3..5 steps of 3 ms do not give it")
    ap.add_argument("--no-kernels", action="store_true",
                    help="skip the per-kernel sections (fields gather, cfg1_protocol, cfg3_autorec, cfg4_hybrid; N = 1 only)")
    ap.add_argument("--kernel-steps", type=int, default=10)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with N > 1 and no torchrun environment
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_argv(argv, gpus, port):
    """Command line of the child that runs the N ranks: the driver's own launch form around this script's arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _descendants(pid):
    """Every live descendant of pid, by walking /proc (torch elastic starts its workers in sessions of their own, so they are
    NOT in the agent's process group and a killpg on the agent does not reach them)."""
    parent = {}
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                with open("/proc/%s/stat" % d) as f:
                    parent[int(d)] = int(f.read().rsplit(")", 1)[1].split()[1])
            except (OSError, ValueError, IndexError):
                pass
    out, frontier = [], [pid]
    while frontier:
        frontier = [c for c, pp in parent.items() if pp in frontier]
        out += frontier
    return out


def _alive(pid):
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


def _run_children(cmd, env, timeout_s, grace_s=10.0):
    """Runs cmd in its own process group; returns (exit code or None on timeout, stdout text, pids that survived the clean-up).

    On timeout the agent gets SIGTERM first (its handler ends the workers it started), then, after a grace period, SIGKILL goes
    to the agent's group and to every descendant recorded BEFORE the first signal."""
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
        return proc.returncode, out.decode(errors="replace"), []
    except subprocess.TimeoutExpired:
        kids = _descendants(proc.pid)
        try:
            proc.send_signal(signal.SIGTERM)
            proc.wait(grace_s)
        except (subprocess.TimeoutExpired, ProcessLookupError):
            pass
        try:
            os.killpg(proc.pid, signal.SIGKILL)          # exactly the group started above
        except ProcessLookupError:
            pass
        for pid in kids:
            try:
                os.kill(pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        out, _ = proc.communicate()
        deadline = time.time() + 5.0
        while time.time() < deadline and any(_alive(p) for p in kids):
            time.sleep(0.1)
        return None, out.decode(errors="replace"), [p for p in kids if _alive(p)]


def _result_line(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                json.loads(line)
                return line
            except ValueError:
                continue
    return None


def visible_gpus():
    """GPUs this process may use, WITHOUT opening the HIP runtime: the *_VISIBLE_DEVICES lists, else the KFD topology (nodes with
    SIMDs are GPUs); torch.cuda.device_count() only when neither says anything (it calls hipGetDeviceCount when amdsmi is absent)."""
    counts = []
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            counts.append(len([x for x in v.split(",") if x.strip() != ""]))
    if counts:
        return min(counts)
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(nodes):
            with open(os.path.join(nodes, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        if n:
            return n
    except (OSError, ValueError):
        pass
    return torch.cuda.device_count()


def self_launch(args, argv):
    """Parent of a plain `bench.py --gpus N`: never initialises the GPU, starts the ranks, relays rank 0's JSON line.

    A run whose ranks fail or hang with the overlapped gather is repeated once, in fresh processes, with one blocking
    all-gather per step (RECO_BENCH_SYNC_GATHER=1): a hang cannot be recovered from inside the hung process."""
    visible = visible_gpus()
    if visible < args.gpus and not args.share_gpus:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible (use --share-gpus to rehearse the N-rank path on "
                         "fewer devices)" % (args.gpus, visible))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    if os.environ.get("RECO_BENCH_DRY_LAUNCH") == "1":    # CPU test of the launcher: print the command, start nothing
        print(json.dumps({"launcher_argv": launcher_argv(argv, args.gpus, 29500)}))
        return 0
    attempts = [("overlapped", {})]
    if os.environ.get("RECO_BENCH_SYNC_GATHER", "0") != "1":
        attempts.append(("blocking", {"RECO_BENCH_SYNC_GATHER": "1", "RECO_BENCH_FELL_BACK": "1"}))
    for mode, extra in attempts:
        cmd = launcher_argv(argv, args.gpus, free_port())
        rc, out, survivors = _run_children(cmd, dict(env, **extra), args.launch_timeout)
        line = _result_line(out)
        if survivors:      # ranks that outlived SIGKILL still hold the GPUs: a second attempt next to them would measure nothing
            print("bench.py: rank processes %s survived the clean-up of the %s-gather attempt; not starting another" % (survivors, mode),
                  file=sys.stderr, flush=True)
            return 1
        if rc == 0 and line:
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
            return 0
        print("bench.py: the %d self-started ranks %s with the %s gather%s" % (
            args.gpus, "timed out after %.0f s" % args.launch_timeout if rc is None else "exited with code %s" % rc, mode,
            "; repeating with one blocking all-gather per step" if (mode, extra) != attempts[-1] else ""), file=sys.stderr, flush=True)
    return 1


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def build_model(n_users, device):
    from recommender_amd.helpers import NCF
    torch.manual_seed(0)
    m = NCF(n_users, N_ITEMS, FACTOR, LAYERS, 0.0, "NeuMF-end").eval()
    with torch.no_grad():                       # N(0, 0.01) init gives scores ~1e-3; scale to trained-model magnitudes
        for name, p in m.named_parameters():
            if "embed" in name:
                p.mul_(25.0)
    return m.to(device)


def cpu_baseline(model, check_ids, check_vals, budget_s=12.0, budget_1t_s=5.0):
    """Reference PyTorch CPU forward (oracle/torch_ref.py: the same ATen ops) on a bounded user sample."""
    from oracle import torch_ref as T
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    threads = torch.get_num_threads()
    chunk = 512
    # parity of the baseline itself against the GPU result, on the first chunk
    users = torch.arange(chunk)
    ids, vals = T.ncf_score_topk(sd, users, K, chunk)
    ok = bool(np.allclose(vals.numpy(), check_vals[:chunk], rtol=1e-5, atol=1e-6))
    same_ids = float((ids.numpy() == check_ids[:chunk]).mean())
    done, t0 = 0, time.perf_counter()
    while True:
        s = done % (N_USERS - chunk)
        T.ncf_score_topk(sd, torch.arange(s, s + chunk), K, chunk)
        done += chunk
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    # the same loop on one thread (SURVEY 8d asks for both)
    torch.set_num_threads(1)
    done1, t1 = 0, time.perf_counter()
    while True:
        s = done1 % (N_USERS - chunk)
        T.ncf_score_topk(sd, torch.arange(s, s + chunk), K, chunk)
        done1 += chunk
        el1 = time.perf_counter() - t1
        if el1 >= budget_1t_s:
            break
    torch.set_num_threads(threads)
    return {"value": done * N_ITEMS / el, "unit": "pairs/s", "cores": threads, "kind": "port",
            "value_1_thread": done1 * N_ITEMS / el1,
            "sample": "%d users x %d items (chunks of %d users x catalog, score + top-%d) in %.1f s; torch %s CPU, "
                      "%d intra-op threads; matches GPU top-%d scores: %s, ids equal %.4f" % (
                          done, N_ITEMS, chunk, K, el, torch.__version__, threads, K, ok, same_ids)}


def measured_traffic(kernel_substr):
    """(HBM bytes per launch of the dominant kernel, file it was read from): FETCH_SIZE x2 + WRITE_SIZE from the committed
    rocprofv3 --pmc passes of this same command (the guide's gfx950 correction).  PMC counters need the profiler: they cannot be
    read from inside the run, so the figure is labelled with its source."""
    best = (None, None)
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith("_hbm_traffic.json"):
            for name, v in json.load(open(os.path.join(pdir, f)))["kernels"].items():
                if kernel_substr in name and "hbm_bytes_per_launch_corrected" in v:
                    best = (v["hbm_bytes_per_launch_corrected"], "profiles/" + f)
    return best


def mfma_roofline(kernel, pairs_per_launch, kernel_ms, launches, traffic=None, traffic_source=None, algorithmic_bytes=None):
    """roofline object for a tower kernel: `achieved` / `frac` count the FLOPs the kernel EXECUTES (hardware utilisation, <= 1);
    the reference's algorithmic 86,176 FLOP/pair (layer 1 is factorised away) sits under `algorithmic`."""
    executed = pairs_per_launch * FLOP_PER_PAIR_EXEC / (kernel_ms * 1e-3) / 1e12
    algorithmic = pairs_per_launch * FLOP_PER_PAIR / (kernel_ms * 1e-3) / 1e12
    r = {"bound": "mfma", "kernel": kernel, "achieved": executed, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
         "frac": executed / PEAK_F32_MFMA_TFLOPS, "flop_per_pair_executed": FLOP_PER_PAIR_EXEC,
         "kernel_ms": kernel_ms, "launches_timed": launches, "pairs_per_launch": pairs_per_launch,
         "algorithmic": {"flop_per_pair": FLOP_PER_PAIR, "tflops": algorithmic, "ratio_to_peak": algorithmic / PEAK_F32_MFMA_TFLOPS,
                         "note": "the reference's FLOP count per pair over the fp32 MFMA peak: a speed-up-inclusive ratio (layer 1 is "
                                 "factorised, W1[u;i] = W1u u + W1i i), NOT a hardware fraction"},
         "traffic": traffic}
    if traffic is not None:
        r["traffic_source"] = traffic_source
        r["traffic_note"] = "HBM bytes per launch from separate rocprofv3 --pmc passes of this command, not live"
        if algorithmic_bytes:
            r["traffic_algorithmic_bytes"] = algorithmic_bytes
    return r


def parallelism_text(world, share_gpus, pipelined, reserve, depth=2):
    """What the ranks actually do with the top-k lists, in words (config.parallelism)."""
    if world == 1:
        from recommender_amd.dist import force_collective_setting
        if force_collective_setting() and dist.is_initialized():
            return ("one rank, REHEARSAL of the RCCL leg (RECO_FORCE_GATHER=1): the pipelined all-gather of packed top-k is issued "
                    "in a one-rank group, every step")
        return "one rank: all users on this GPU, no collective is issued (user-sharded with replicated tables at N > 1)"
    backend = "gloo (REHEARSAL: ranks share GPUs, numbers meaningless)" if share_gpus else "RCCL"
    if not pipelined:
        how = ("one blocking all-gather per step; the pipelined form failed and the ranks were restarted" if os.environ.get("RECO_BENCH_FELL_BACK") == "1"
               else "one blocking all-gather per step (RECO_BENCH_SYNC_GATHER=1)")
    elif reserve:
        how = ("pipelined: the gather of step i is enqueued behind step i's top-k and awaited before step i + %d; the tile kernels leave "
               "%d of the CUs free (RECO_RESERVE_CUS) so that the gather kernel can run under step i + 1's tile launch" % (depth, reserve))
    else:
        how = ("pipelined, %d buffers in flight: the gather of step i is enqueued behind step i's top-k and awaited before step i + %d; "
               "the persistent tile kernel fills every CU's LDS, so the gather kernel may only find CUs in the gap between two tile "
               "launches (beside the select kernels) -- with three buffers it has had a whole step by the time it is awaited; "
               "RECO_RESERVE_CUS=r would keep r CUs free for it" % (depth, depth))
    return "user-sharded x%d, replicated tables, %s all-gather of packed top-k (%s)" % (world, backend, how)


def main_rank(args):
    # stdout carries exactly ONE line, the result: everything else that writes to file descriptor 1 in this process (RCCL prints
    # a version banner there at communicator creation) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before anything opens the HIP runtime (RCCL needs dmabuf IPC on this pool)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # under torch.distributed.run (also at N = 1)
    n_dev = torch.cuda.device_count()
    if args.share_gpus:
        local = local % max(n_dev, 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = "gloo" if args.share_gpus else "nccl"
    if launched:
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a stuck collective ends the ranks (RCCL's watchdog) instead of hanging them; the parent then repeats with the blocking gather
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=180), **kw)

    from recommender_amd import _lib, ops
    from recommender_amd.dist import TopkGather, shard_range

    total_users = N_USERS * world
    model = build_model(total_users, device)
    start, count = shard_range(total_users, rank, world)
    assert count == N_USERS
    L = _lib.lib()

    def fence():
        torch.cuda.synchronize(device)
        if launched:
            dist.barrier()
        torch.cuda.synchronize(device)

    class _SyncGather:
        """Same interface as TopkGather without the overlap: one blocking all-gather per step (fallback only)."""

        def __init__(self):
            self.last = None

        def step(self):
            from recommender_amd.dist import sharded_score_topk
            done, self.last = self.last, sharded_score_topk(model, K, total_users)
            return done

        def flush(self):
            done, self.last = self.last, None
            return done

    pipelined = os.environ.get("RECO_BENCH_SYNC_GATHER", "0") != "1"
    from recommender_amd.dist import reserved_cus_setting
    reserve = reserved_cus_setting() if world > 1 else 0
    # buffers in flight (dist.TopkGather): with peers a third buffer lets a gather that only found CUs when the NEXT tile launch ended
    # complete before it is awaited; alone it changes nothing (profiles/r03_force_gather_n1.jsonl)
    depth = int(os.environ.get("RECO_GATHER_DEPTH", "3" if world > 1 else "2"))

    spread = {}                                   # per-launch spread of the tile kernel's duration, by precision

    def timed_run(precision):
        """W warm-up steps, then K timed steps between fences; (seconds max over ranks, mean ms of the tile kernel, ids, vals)."""
        ops.set_tower_precision(precision)
        # step i's all-gather is enqueued behind step i's kernels and not waited for before step i + 2 (N > 1; one rank gathers nothing)
        pipe = TopkGather(model, K, total_users, depth=depth) if pipelined else _SyncGather()
        fence()                                  # first barrier of the process group (communicator set-up) outside everything timed
        # The device clock needs ~30 ms of load to settle after idle (a tile launch runs 3.6 ms cold, 3.13 ms settled), and any
        # host-side pause -- a process group's first barrier -- sends it back: `--precondition` untimed steps (the SAME fixed
        # count on every rank) come before the W warm-up steps, so that the K timed steps see the sustained clock.
        for _ in range(max(args.precondition, 0) + args.warmup):
            pipe.step()
        pipe.flush()
        cap = args.steps + 4                     # the tile kernel's launches only: an event pair costs ~4 us of stream time
        _lib.check(L.reco_profile_enable_kinds(cap, 1 << _lib.PROF_TOWER))
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.step()
        ids, vals = pipe.flush()
        fence()
        el = time.perf_counter() - t0
        ms = (ctypes.c_float * cap)()
        n = ctypes.c_int(0)
        _lib.check(L.reco_profile_collect(ms, cap, ctypes.byref(n)))
        _lib.check(L.reco_profile_enable(0))
        per_launch = np.asarray(list(ms)[: n.value], np.float64)
        kernel_ms = float(per_launch.mean()) if n.value else float("nan")
        spread[precision] = {"median": float(np.median(per_launch)), "min": float(per_launch.min()), "max": float(per_launch.max())} if n.value else None
        if os.environ.get("RECO_BENCH_DEBUG") == "1":
            print("tile kernel ms per launch (%s): %s" % (precision, " ".join("%.3f" % x for x in list(ms)[: n.value])), file=sys.stderr, flush=True)
        t = torch.tensor([el], dtype=torch.float64, device=device)
        if launched:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), kernel_ms, n.value, ids.clone(), vals.clone()

    # the headline: exact fp32 MFMA (every product the exact fp32 product).  No in-process fallback: a gather that hangs cannot be
    # recovered from here (GatherTimeout / the watchdog end the rank; self_launch() repeats the run with the blocking gather).
    elapsed, tile_ms, n_timed, ids, vals = timed_run("f32")
    split = None
    if not args.no_split:      # the same job in the opt-in split-operand mode (fp32 = 3 bf16 pieces on the bf16 MFMA, fp32 accumulate)
        split = timed_run("bf16x3")
        ops.set_tower_precision("f32")

    # second half of BASELINE.json's metric: full-catalog top-100 users/s on the synthetic 10M x 1M NeuMF (configs[4]), timed on
    # a slice of 1,024 users per GPU per step, after the cfg2 region
    cfg5 = None
    if not args.no_cfg5:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_cfg5
        cfg5 = bench_cfg5.run(device, world, rank, launched, args.cfg5_users_per_gpu, args.cfg5_steps, 1, "f32")
        if not args.no_split:
            s5 = bench_cfg5.run(device, world, rank, launched, args.cfg5_users_per_gpu, args.cfg5_steps, 1, "bf16x3")
            if rank == 0:
                cfg5["split_bf16x3"] = {k: s5[k] for k in ("value", "unit", "pairs_per_s", "ms_per_step", "roofline")}

    kernels = None
    if world == 1 and not args.no_kernels:       # the other roofline targets of north_star (one GPU is their whole story: SURVEY 8e)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_kernels
        kernels = bench_kernels.run(model, device, args.kernel_steps, 3)

    if rank == 0:
        pairs_per_rank = N_USERS * N_ITEMS
        pairs = pairs_per_rank * world * args.steps
        traffic, traffic_src = measured_traffic("tower_tile_kernel<32, true") if world == 1 else (None, None)
        res = {
            "metric": "scored user-item pairs/sec (NeuMF fwd, whole node); full-catalog top-100 users/s",
            "value": pairs / elapsed, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "users_per_s_top%d" % K: N_USERS * world * args.steps / elapsed,
            "config": {"workload": "cfg2: NeuMF (factor 32, 3 MLP layers) full-catalog scoring + top-10, ML-1M NCF id space "
                                   "6038 users x 3533 items = 21,332,254 pairs per GPU per step (`value`); cfg5 (10M x 1M, top-100) in "
                                   "`cfg5_top100`, same ranks, same protocol",
                       "users_per_gpu": N_USERS, "items": N_ITEMS, "k": K, "weights": "random init, reference ctor order, seed 0, embeddings x25",
                       "precondition_steps": max(args.precondition, 0),
                       "layer1_projections": "P[u] = W1u eu + b1 and Q[i] = W1i ei are weight-only quantities the handle computes once per weight "
                                             "version at creation (as forward(user, item) always read them); a step is the tile kernel + the select "
                                             "(until late in round 3 every step recomputed them for its user range: two 9-us GEMM launches)",
                       "parallelism": parallelism_text(world, args.share_gpus, pipelined, reserve, depth)},
            "roofline": mfma_roofline("tower_tile_kernel<32,true>", pairs_per_rank, tile_ms, n_timed, traffic, traffic_src,
                                      algorithmic_bytes=pairs_per_rank * 4),
        }
        if spread.get("f32"):
            res["roofline"]["kernel_ms_per_launch"] = spread["f32"]
        if split is not None:
            s_elapsed, s_tile_ms, _, s_ids, s_vals = split
            dev = (s_vals - vals).abs() / vals.abs().clamp_min(1e-30)
            res["split_bf16x3"] = {
                "value": pairs / s_elapsed, "unit": "pairs/s", "ms_per_step": s_elapsed / args.steps * 1e3,
                "kernel": "tower_tile_split_kernel<32,true>", "kernel_ms": s_tile_ms, "speedup_vs_f32": elapsed / s_elapsed,
                "top%d_ids_equal_to_f32" % K: float((s_ids == ids).double().mean().item()),
                "top%d_scores_max_rel_dev_vs_f32" % K: float(dev.max().item()),
                "note": "opt-in RECO_PREC_SPLIT_BF16X3 (not the headline): fp32 operands split exactly into three bf16 pieces, six cross "
                        "terms on v_mfma_f32_32x32x16_bf16, fp32 accumulate; passes the same parity tests as the exact mode"}
        if cfg5 is not None:
            if world == 1 and cfg5.get("roofline"):
                t5, t5_src = measured_traffic("tower_tile_topk_kernel<32, true")
                if t5 is not None:
                    cfg5["roofline"].update({"traffic": t5, "traffic_source": t5_src,
                                             "traffic_note": "HBM bytes per launch from separate rocprofv3 --pmc passes of this command, not live",
                                             "traffic_algorithmic_bytes": args.cfg5_users_per_gpu * 100 * 8 + (10 ** 6) * (128 + 32) * 4})
            res["cfg5_top100"] = cfg5
        if kernels is not None:
            res.update(kernels)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(model, ids.cpu().numpy(), vals.cpu().numpy())
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    os.close(result_fd)
    if launched:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args, argv)
    return main_rank(args)


if __name__ == "__main__":
    sys.exit(main())
