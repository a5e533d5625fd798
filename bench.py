#!/usr/bin/env python3
"""Headline benchmark: sampled edges/sec of neighbor_sampling_homogenous, fanout [15,10], batch 1024,
on a synthetic RMAT scale-24 graph (BASELINE.json configs[1]), HIP path through the C ABI.

A "step" is one pass of the hot path over one batch of synthetic input: ONE launch of tg_ns_homo_batched over
`--batches-per-step` independent 1024-seed mini-batches (seeds, CSC and output slabs resident in HBM).  The
throughput therefore does not depend on how many steps the caller asks for.
N > 1: one process per GPU (torchrun), CSC replicated, every rank samples its own K steps with no
data-path collective (weak scaling); the only collectives are the barrier / MAX / SUM of the timing
protocol.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tch-geometric_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--fanout", type=str, default="15,10")
    ap.add_argument("--batches-per-step", "--batches-per-launch", dest="batches_per_step", type=int, default=16384,
                    help="independent 1024-seed mini-batches sampled by the one launch of a step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of BASELINE cfg3 / cfg4")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the (untimed) check of batches of the last timed step against the fused kernel and the oracle")
    ap.add_argument("--idx32", type=int, default=1, help="also keep a u32 shadow of `indices` for the gathers")
    ap.add_argument("--ptr32", type=int, default=1, help="also keep a u32 shadow of `ptrs`")
    ap.add_argument("--placements", type=int, default=2,
                    help="sets of output slabs + workspace to allocate at start-up; the launch is timed once on each and "
                         "the fastest set is kept, the others freed (where an allocation lands decides 5-10 %% of the "
                         "launch time, DESIGN.md 4.1b); 1 = take the first")
    ap.add_argument("--replies", choices=["auto", "compact"], default="auto",
                    help="--mode partitioned: auto = fixed-size slot replies where they apply (one size read-back per hop); "
                         "compact = counts + entries (two read-backs per hop), the round-2/3 protocol")
    ap.add_argument("--mode", choices=["replicated", "partitioned"], default="replicated",
                    help="replicated: CSC on every rank, seed batches sharded (the headline).  partitioned: every rank owns "
                         "the columns of a contiguous vertex range; remote neighbours are fetched by all-to-all (cfg5 shape)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="--mode partitioned with one rank: still run every collective of the protocol over RCCL (the rank "
                         "exchanges with itself) -- what a one-GPU box can exercise of the transport")
    ap.add_argument("--lanes", type=int, default=1,
                    help="--mode partitioned: super-batches in flight at once -- each lane a PartitionedSampler of its own on "
                         "its own HIP stream, all driven from ONE host thread over ONE communicator in a fixed, rank-independent "
                         "enqueue order (tch_geometric.partitioned.interleave), so that the exchange of one super-batch overlaps "
                         "the sampling of the next (SURVEY.md 8(e)); legal with any number of ranks")
    ap.add_argument("--pipelines", choices=["auto", "push", "staged", "staged2"], default="auto",
                    help="pipeline of the window-ordered launch: auto = time all of them in the untimed set-up and keep the "
                         "fastest (staged2 = the staged pipeline in two parts, the emit pass of one beside the chain of the next)")
    ap.add_argument("--form", choices=["auto", "windowed", "fused"], default="auto",
                    help="tg_ns_homo_batched_ws form: window-ordered gather of the launch, or the fused per-batch kernel")
    return ap.parse_args(argv)


_RESULT_FD = None


def keep_stdout_for_the_result():
    """The contract is ONE JSON line on stdout.  RCCL prints its version banner to stdout when a communicator starts, so
    file descriptor 1 is pointed at stderr for the run and the result line is written to the saved descriptor."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


def print_result(result):
    line = (json.dumps(result) + "\n").encode()
    sys.stdout.flush()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, line)


def self_launch_command(argv, n_gpus, port):
    """`python bench.py --gpus N` without a launcher around it: the command this process starts as a CHILD (one rank per
    GPU over RCCL; the contract's own spelling).  127.0.0.1: the container hostname may not resolve."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def needs_self_launch(args, environ):
    return args.gpus > 1 and "WORLD_SIZE" not in environ


def self_launch(args, argv):
    """Runs before torch is imported and before anything touches the GPU: starts torchrun as a child process (never an
    exec), lets its stderr through, relays the ONE JSON line rank 0 printed and exits with the child's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.run(self_launch_command(argv, args.gpus, port), env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{") and "\"metric\"" in ln]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif proc.returncode == 0:
        sys.stderr.write("bench.py: the launched ranks printed no result line\n")
        raise SystemExit(1)
    raise SystemExit(proc.returncode)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if needs_self_launch(args, os.environ):
        return self_launch(args, argv)
    keep_stdout_for_the_result()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    import torch
    import torch.distributed as dist

    from tch_geometric import _cabi  # raises if the gfx950 library is missing (no fallback)
    from tch_geometric import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # TG_BENCH_REHEARSE=1: rehearsal of the N > 1 path on a one-GPU box -- every rank uses cuda:0 and the timing
    # protocol's collectives go over gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearse = os.environ.get("TG_BENCH_REHEARSE", "0") == "1"
    dev = torch.device("cuda", 0 if rehearse else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    fanout = [int(x) for x in args.fanout.split(",")]
    n_nodes = 1 << args.scale
    n_edges = n_nodes * args.edge_factor
    K, W, G, B = args.steps, args.warmup, args.batches_per_step, args.batch
    if args.mode == "partitioned":
        return partitioned_mode(args, torch, dist, _cabi, sharding, dev, world, rank, fanout)

    form = {"auto": 0, "windowed": 1, "fused": 2}[args.form]

    ws_staged = form != 2 and args.pipelines != "push"   # the workspaces then hold the staged pipeline's stage slots too
    # the arenas are allocated BEFORE the graph exists (placement), so the stage slots are sized for an
    # ASSUMED bound on the longest column (edges / 256: one 64-byte chunk per frontier vertex on RMAT-24); a graph beyond
    # the bound needs two chunks, finds the workspace too small for them and takes the push pipeline (reported as such)
    ws_sizing = _cabi.graph_sizing(n_nodes, n_edges, max(n_edges >> 8, 1))

    def alloc_slabs(G):  # 16 384 batches per launch need ~100 GB of slabs + workspace: on a GPU with less free HBM, halve
        while True:
            try:
                out = _cabi.NsBatchedOut(G, B, fanout, dev)
                ws = _cabi.ns_homo_workspace(G, B, fanout, dev, staged=ws_staged, graph=ws_sizing) if form != 2 else None
                return out, ws, G
            except torch.OutOfMemoryError:
                out = ws = None
                torch.cuda.empty_cache()
                if G <= 256:
                    raise
                G //= 2

    # the big arenas first, while the device's memory is still one piece (TG_BENCH_SLABS_FIRST=0: after the graph build)
    slabs_first = os.environ.get("TG_BENCH_SLABS_FIRST", "1") == "1"
    candidates = []
    if slabs_first:
        out, ws, G = alloc_slabs(G)
        candidates.append((out, ws))
        for _ in range(max(args.placements, 1) - 1):  # further placements of the same arenas, as long as they fit
            try:
                free_b, _total = torch.cuda.mem_get_info(dev)
                need = sum(t.numel() * 8 for t in (out.samples, out.rows, out.cols, out.edge_index)) + \
                    (ws.numel() * 8 if ws is not None else 0)
                if free_b < need + (24 << 30):           # keep room for the graph, its build and the seeds
                    break
                candidates.append((_cabi.NsBatchedOut(G, B, fanout, dev),
                                   _cabi.ns_homo_workspace(G, B, fanout, dev, staged=ws_staged, graph=ws_sizing)
                                   if form != 2 else None))
            except torch.OutOfMemoryError:
                torch.cuda.empty_cache()
                break

    # ---- graph: R-MAT edges -> CSC with the reference's sort key (storage.rs:118-123); resident in HBM
    t_build = time.time()
    row, col = _cabi.rmat_edges(args.scale, n_edges, 0x5EED0000 + args.scale, dev)
    ptrs, indices, perm = _cabi.coo_to_csx(row, col, n_nodes, n_nodes, True)
    del row, col, perm
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t_build = time.time() - t_build
    # u32 shadows (int32 tensors carrying the u32 bit pattern): only while ids / offsets fit 32 bits
    idx32 = indices.to(torch.int32) if args.idx32 and n_nodes <= 2 ** 32 else None
    ptr32 = ptrs.to(torch.int32) if args.ptr32 and n_edges < 2 ** 32 else None
    graph = _cabi.graph_view(ptrs, indices, indices32=idx32, ptrs32=ptr32, max_degree="auto")
    if not slabs_first:
        out, ws, G = alloc_slabs(G)
        candidates = [(out, ws)]

    # ---- this rank's mini-batches: global batch ids [rank*(W+K)*G, (rank+1)*(W+K)*G); step i samples G of them with
    # call ids first + i*G ...  Seeds of up to 32 steps are kept resident (268 MB each); longer runs cycle through them
    # with fresh call ids, i.e. fresh draws.
    first, _ = sharding.rank_batch_range(rank, world, (W + K) * G)
    n_pool = max(1, min(W + K, 32))
    seeds = _cabi.seed_batches(0xBA7C4, first, n_pool * G, B, n_nodes, dev)
    placement_ms = None
    in_flight = None
    PIPELINE_KNOBS = {"push": dict(staged=0), "staged": dict(staged=1, stage_parts=1), "staged2": dict(staged=1, stage_parts=2)}
    pipelines = ["push"]
    if form != 2 and args.pipelines != "push":
        # the window-ordered launch has two pipelines with identical outputs (push: emit -> sort -> gather scattering its
        # samples; staged: two-level sort -> gather into 64-byte stage slots -> emit all four streams as pure streams);
        # which is faster depends on where the slabs landed, so the set-up times them, like a planner would
        pipelines = ["push", "staged", "staged2"] if args.pipelines == "auto" else [args.pipelines]
    pipeline = pipelines[0]
    _cabi.ns_win_tuning_set(**PIPELINE_KNOBS[pipeline])
    if len(candidates) > 1 or len(pipelines) > 1:
        # untimed set-up: the launch is timed on every combination of {samples slab} x {rows / cols / edge_index slabs} x
        # {workspace} of the placements (the gather kernel's time follows the first, the emit kernel's the second), one
        # warm + two timed launches each; the fastest combination is kept, the rest freed
        import copy
        import itertools
        placement_ms = {}
        best, best_ms = None, None
        n_c = len(candidates)
        for (a, b_, c), pl in itertools.product(itertools.product(range(n_c), range(n_c), range(n_c) if form != 2 else [0]),
                                                pipelines):
            _cabi.ns_win_tuning_set(**PIPELINE_KNOBS[pl])
            mix = copy.copy(candidates[a][0])
            mix.samples = candidates[a][0].samples
            mix.rows, mix.cols, mix.edge_index = (candidates[b_][0].rows, candidates[b_][0].cols,
                                                  candidates[b_][0].edge_index)
            w_c = candidates[c][1]
            _cabi.ns_homo_batched(graph, seeds[:G], fanout, 0, first, mix, ws=w_c, form=form)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(2):
                _cabi.ns_homo_batched(graph, seeds[:G], fanout, 0, first, mix, ws=w_c, form=form)
            ev1.record()
            torch.cuda.synchronize(dev)
            ms = ev0.elapsed_time(ev1) / 2
            placement_ms["samples%d_streams%d_ws%d_%s" % (a, b_, c, pl)] = round(ms, 3)
            if best_ms is None or ms < best_ms:
                best, best_ms, pipeline = (a, b_, c), ms, pl
        _cabi.ns_win_tuning_set(**PIPELINE_KNOBS[pipeline])
        # beside the line (never part of `value`): the same launches with TWO in flight, one per placement on its own
        # stream -- what a caller that prefetches the next super-batch gets
        in_flight = None
        if n_c >= 2:
            side = [torch.cuda.Stream(device=dev) for _ in range(2)]
            cur = torch.cuda.current_stream(dev)
            for timed in (False, True):
                torch.cuda.synchronize(dev)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for st_ in side:
                    st_.wait_stream(cur)
                for i in range(8):
                    with torch.cuda.stream(side[i % 2]):
                        _cabi.ns_homo_batched(graph, seeds[:G], fanout, 0, first + i * G, candidates[i % 2][0],
                                              ws=candidates[i % 2][1], form=form)
                for st_ in side:
                    cur.wait_stream(st_)
                ev1.record()
                torch.cuda.synchronize(dev)
                if timed:
                    in_flight = {"launches": 8, "ms_per_launch": round(ev0.elapsed_time(ev1) / 8, 3)}
        out = copy.copy(candidates[best[0]][0])
        out.samples = candidates[best[0]][0].samples
        out.rows, out.cols, out.edge_index = (candidates[best[1]][0].rows, candidates[best[1]][0].cols,
                                              candidates[best[1]][0].edge_index)
        ws = candidates[best[2]][1]
        mix = w_c = None
        candidates = None
        torch.cuda.empty_cache()
    acc = torch.zeros(3, dtype=torch.int64, device=dev)  # sampled edges, frontier slots, launches

    def run(lo, hi, events=None):
        for i in range(lo, hi):
            s0 = (i % n_pool) * G
            if events is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            _cabi.ns_homo_batched(graph, seeds[s0:s0 + G], fanout, 0, first + i * G, out, ws=ws, form=form)
            if events is not None:
                ev1.record()
                events.append((ev0, ev1))
            ne = out.counts[:, 1].sum()
            # frontier slots expanded = seeds + samples that existed when the last hop started
            nf = out.layer_offsets[:, len(fanout) - 1, 0].sum() if fanout else torch.zeros((), dtype=torch.int64, device=dev)
            acc.add_(torch.stack([ne, nf, torch.ones((), dtype=torch.int64, device=dev)]))

    def fence():
        sharding.fence(dev)

    run(0, W)
    acc.zero_()
    events = []
    fence()
    gpu_state0 = gpu_state(local_rank if not rehearse else 0)
    t0 = time.perf_counter()
    run(W, W + K, events)
    fence()
    dt = time.perf_counter() - t0
    gpu_state1 = gpu_state(local_rank if not rehearse else 0)

    dt_max, tot = sharding.reduce_measurement(dt, acc)
    edges_all = int(tot.tolist()[0])

    # ---- roofline of the dominant kernel (this rank): algorithmic bytes / HIP-event kernel time
    kernel_ms = [a.elapsed_time(b) for a, b in events]
    avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    my_edges, my_frontier, my_launches = (int(x) for x in acc.tolist())
    # per hop: 24 B per frontier slot (8 id + 16 ptrs pair) + 40 B per sampled edge (8 gather + 32 written);
    # per batch: 16 B per seed (read + copy into `samples`)            -- SURVEY.md 8(d)
    alg_bytes = 24 * my_frontier + 40 * my_edges + 16 * B * G * K
    bytes_per_launch = alg_bytes / my_launches
    achieved = bytes_per_launch / avg_kernel_s / 1e9
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % (pipeline.rstrip("2") if form != 2 else "fused"))
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))  # PMC passes of an identical launch, collected by tools/pmc_win.sh
            form_ran = "fused" if args.form == "fused" else "windowed"
            if (tj.get("batches_per_launch") == G and args.scale == 24 and B == 1024 and fanout == [15, 10]
                    and tj.get("idx32") == args.idx32 and tj.get("ptr32") == args.ptr32 and tj.get("form") == form_ran
                    and (form_ran == "fused" or tj.get("pipeline", "push") == pipeline.rstrip("2"))):
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = ("profiles/%s (separate rocprofv3 --pmc passes of the same launch and pipeline; NOT this "
                                  "run)" % os.path.basename(tpath))
        except Exception:
            traffic = None

    # what the launches really took (a query of the library, not the flag that was asked for)
    if ws is None or _cabi.ns_homo_batched_form(graph, out, G, B, fanout, ws=ws, form=form)[0] == 2:
        pipeline_taken = "fused"
    else:
        pipeline_taken = "push"
        if _cabi.ns_homo_batched_staged(graph, out, G, B, fanout, ws=ws, form=form):
            pipeline_taken = "staged, %d part(s)" % _cabi.ns_win_tuning()["stage_parts"]

    result = {
        "metric": "sampled edges/sec, neighbor_sampling_homogenous fanout [%s] on RMAT-%d" % (args.fanout, args.scale),
        "value": edges_all / dt_max,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt_max / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {
            "workload": "neighbor_sampling_homogenous, RMAT scale-%d (%d nodes / %d edges, CSC i64), fanout %s, "
                        "batch %d, default sampler (uniform w/o replacement), no filter" %
                        (args.scale, n_nodes, n_edges, fanout, B),
            "step": "one launch over %d independent %d-seed mini-batches" % (G, B),
            "placement_policy": ("the launch is timed on the FASTEST of the slab / workspace placements x pipelines tried in "
                                 "untimed set-up (all listed in placements_tried_ms_per_launch; --placements 1 "
                                 "--pipelines push = take the first)" if placement_ms else "first allocation"),
            "pipeline": pipeline_taken,
            "first_placement_ms_per_launch": (next(iter(placement_ms.values())) if placement_ms else None),
            "batches_per_launch": G,
            "placements_tried_ms_per_launch": placement_ms,
            "two_launches_in_flight": in_flight,
            "hbm_layout": "CSC int64 ptrs/indices%s%s" % (" + u32 shadow of indices for the gathers" if args.idx32 else "",
                                                          " + u32 shadow of ptrs" if args.ptr32 else ""),
            "rng": "philox4x32-10 counter-addressed (slot draws: one 32-bit word each, Lemire's exact rejection, 64-bit fallback), "
                   "seed 0, call_id = global batch id",
            "max_degree": graph.max_degree,
            "parallelism": "replicated CSC, %d x independent seed batches" % world,
            "sampled_edges_per_step": edges_all / (K * world),
            "sampled_edges_per_mini_batch": edges_all / (K * world * G),
            "graph_build_s": round(t_build, 2),
            "form": args.form,
            "gpu_state_before_timed_region": gpu_state0,
            "gpu_state_after_timed_region": gpu_state1,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": ("ns_homo_uniform_kernel (fused per-batch form)" if args.form == "fused" else
                       "tg_ns_homo_batched_ws launch, window-ordered form, pipeline `%s` (push: first hops + emit -> counting "
                       "sort of the frontier by window -> win_gather; staged: first hop -> two-level sort -> gather into stage "
                       "slots -> emit); no kernel dominates: duration = the whole launch" % pipeline_taken),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "avg_launch_ms": avg_kernel_s * 1e3,
            "launches": my_launches,
        },
    }

    # ---- what was timed is checked (untimed): batches of the LAST timed step, as they lie in the slabs, against the
    # fused per-batch kernel (another code path, on the device) and against the CPU oracle in philox-mode (the checker)
    if not args.no_verify:
        i_last = W + K - 1
        result["config"]["verified_batches"] = verify_step(
            torch, _cabi, graph, ptrs, indices, out, seeds[(i_last % n_pool) * G:(i_last % n_pool) * G + G], fanout,
            first + i_last * G, sorted({0, G // 2, G - 1}), oracle=(rank == 0))

    if rank == 0 and world == 1 and not args.no_secondary:
        del out, ws
        torch.cuda.empty_cache()
        result["secondary"] = secondary_configs(torch, _cabi, dev, args.cpu_seconds)
        try:  # the headline launch at the sizes a loader uses (same graph, same seeds rule)
            result["secondary"]["batches_per_launch_sweep"] = launch_size_sweep(torch, _cabi, graph, dev, B, fanout, n_nodes, first)
        except Exception as e:  # noqa: BLE001  (a secondary run must never cost the headline line)
            result["secondary"]["batches_per_launch_sweep"] = {"error": repr(e)}
        try:  # mode 2 (range-partitioned CSC) with this one rank owning everything: the protocol's own cost
            result["secondary"]["mode2_partitioned_one_rank_4096_batches"] = mode2_one_rank(
                torch, _cabi, ptrs, indices, dev, B, fanout, n_nodes, first)
        except Exception as e:  # noqa: BLE001
            result["secondary"]["mode2_partitioned_one_rank_4096_batches"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, ptrs, indices, seeds[:min(int(seeds.shape[0]), 16384)], fanout)
    if rank == 0:
        print_result(result)
    if world > 1:
        dist.destroy_process_group()


def verify_step(torch, _cabi, graph, ptrs, indices, out, step_seeds, fanout, call0, batches, oracle=True):
    """Batches of a step as bench.py left them in the slabs == the fused per-batch kernel on the same seeds and call ids ==
    the oracle (philox-mode; neighbor_sampling.rs:188-223).  Raises on any difference; -> what was checked."""
    dev = step_seeds.device
    counts = out.counts.cpu()
    checked = []
    hp = hi = None
    for j in batches:
        got = out.batch(j, counts)
        one = _cabi.NsBatchedOut(1, step_seeds.shape[1], fanout, dev)
        _cabi.ns_homo_batched(graph, step_seeds[j:j + 1].contiguous(), fanout, 0, call0 + j, one, form=2)
        ref = one.batch(0)
        if got[4] != ref[4] or not all(torch.equal(a, b) for a, b in zip(got[:4], ref[:4])):
            raise SystemExit("bench.py: batch %d of the last timed step differs from the fused kernel" % j)
        entry = {"batch": j, "call_id": call0 + j, "sampled_edges": int(counts[j][1]), "equals": ["fused kernel"]}
        if oracle:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import numpy as np
            import orc  # the checker
            if hp is None:
                hp, hi = ptrs.cpu().numpy(), indices.cpu().numpy()
            o = orc.ns_homo(hp, hi, step_seeds[j].cpu().numpy(), fanout, orc.rng_philox(0, call0 + j))
            if got[4] != o[4] or not all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(got[:4], o[:4])):
                raise SystemExit("bench.py: batch %d of the last timed step differs from the oracle" % j)
            entry["equals"].append("oracle philox-mode")
        checked.append(entry)
    return checked


def partitioned_mode(args, torch, dist, _cabi, sharding, dev, world, rank, fanout):
    """Mode 2 (BASELINE cfg5 shape): rank r owns the columns of vertices [r*S, (r+1)*S); a step = one
    PartitionedSampler.sample() over G seed batches per rank (requests / replies all-to-all per hop over RCCL; with one
    rank the exchange is skipped and nothing is read back).  Same JSON line as the replicated mode."""
    from tch_geometric import partitioned
    n = 1 << args.scale
    K, W, B = args.steps, args.warmup, args.batch
    G = min(args.batches_per_step, 4096)
    size = partitioned.CscShard.shard_size_for(n, world)
    v_lo, v_hi = min(rank * size, n), min((rank + 1) * size, n)
    t_build = time.time()
    row, col = _cabi.rmat_edges(args.scale, n * args.edge_factor, 0x5EED0000 + args.scale, dev)
    below = int((col < v_lo).sum())         # global edge offset of the shard = edges of the columns before v_lo
    if world > 1:                           # keep the shard's columns (in pieces: masks of 2^31 elements overflow torch's indexing)
        rows_, cols_ = [], []
        for lo in range(0, col.numel(), 1 << 28):
            c = col[lo:lo + (1 << 28)]
            keep = (c >= v_lo) & (c < v_hi)
            rows_.append(row[lo:lo + (1 << 28)][keep])
            cols_.append(c[keep] - v_lo)
        del row, col, keep, c
        row, col = torch.cat(rows_), torch.cat(cols_)
        del rows_, cols_
    ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, v_hi - v_lo, True)
    del row, col
    shard = partitioned.CscShard(ptrs, idx, v_lo, v_hi, below, n, size, n_edges_global=n * args.edge_factor)
    torch.cuda.synchronize()
    t_build = time.time() - t_build
    if args.force_exchange and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    lanes = max(1, args.lanes)
    exchanging = world > 1 or args.force_exchange
    # a communicator per lane (their collectives then pass each other); legal with any number of ranks because ONE host thread
    # enqueues every collective of every lane in a fixed, rank-independent order (partitioned.interleave)
    groups = [dist.new_group(ranks=list(range(world))) for _ in range(lanes)] if (exchanging and lanes > 1) else None
    pipe = partitioned.PipelinedPartitionedSampler(shard, G, B, fanout, lanes=lanes, groups=groups,
                                                   force_exchange=args.force_exchange,
                                                   slot_replies=None if args.replies == "auto" else False)
    ps = pipe.samplers[0]
    firsts = [sharding.rank_batch_range(r, world, (W + K) * G)[0] for r in range(world)]
    first = firsts[rank]
    accs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(lanes)]   # one per lane (= per stream)
    events = []

    def run(lo, hi, timed):
        """super-batches lo .. hi - 1; with --lanes > 1 two (or more) are in flight, interleaved by ONE host thread"""
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        pipe.sample_many(hi - lo,
                         lambda i: _cabi.seed_batches(0xBA7C4, first + (lo + i) * G, G, B, n, dev),
                         0, lambda i: (first + (lo + i) * G, [f + (lo + i) * G for f in firsts]),
                         lambda i, out: accs[i % lanes].add_(out.counts[:, 1].sum()))
        if timed:
            ev1.record()
            events.append((ev0, ev1, hi - lo))

    run(0, W, False)
    torch.cuda.synchronize(dev)
    for a in accs:
        a.zero_()
    sharding.fence(dev)
    t0 = time.perf_counter()
    run(W, W + K, True)
    sharding.fence(dev)
    dt = time.perf_counter() - t0
    acc = torch.stack(accs).sum(0)
    dt_max, tot = sharding.reduce_measurement(dt, torch.cat([acc, torch.zeros(2, dtype=torch.int64, device=dev)]))
    edges_all = int(tot.tolist()[0])
    ms = [a.elapsed_time(b) / cnt for a, b, cnt in events]
    result = {
        "metric": "sampled edges/sec, neighbor_sampling_homogenous fanout [%s] on RMAT-%d" % (args.fanout, args.scale),
        "value": edges_all / dt_max, "unit": "edges/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt_max / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int64", "data": "synthetic",
        "config": {
            "workload": "neighbor_sampling_homogenous, RMAT scale-%d, fanout %s, batch %d, default sampler, no filter; "
                        "CSC RANGE-PARTITIONED over %d rank(s)" % (args.scale, fanout, B, world),
            "step": "one PartitionedSampler.sample() over %d seed batches per rank" % G,
            "batches_per_step_per_rank": G,
            "parallelism": "range-partitioned CSC (contiguous vertex ranges) over %d rank(s); per hop requests and %s "
                           "replies travel by all_to_all_single (RCCL)%s" %
                           (world, "fixed-size slot" if ps.slots else "compact", "" if world > 1 else ("; one rank exchanging with itself over RCCL (all collectives and "
                                                         "read-backs of the multi-rank protocol run)" if args.force_exchange
                                                         else "; one rank: no exchange, no host read-back")),
            "replies": ({"form": "slots", "bytes_per_request_by_hop": [4 * w for w in ps.slot_words],
                         "size_read_backs_per_hop": 1 if exchanging else 0} if ps.slots else
                        {"form": "compact", "bytes_per_entry": 8 * ps.reply_words,
                         "size_read_backs_per_hop": 2 if exchanging else 0}),
            "lanes": lanes,
            "avg_call_ms_this_rank": sum(ms) / len(ms),
            "shard_build_s": round(t_build, 2),
        },
    }
    if rank == 0:
        print_result(result)
    if dist.is_initialized():
        dist.destroy_process_group()


def mode2_one_rank(torch, _cabi, ptrs, indices, dev, B, fanout, n_nodes, first, G=4096, reps=6):
    """SURVEY.md 8(e) mode 2 with ONE rank that owns every column: requests -> owner-side sampling into slot replies -> emit,
    no exchange, nothing read back -- what the partition costs a rank before any link is involved (the replicated launch
    of the same batches is `batches_per_launch_sweep`).  Checked against the replicated launch (counts and edge pointers of
    every batch).  `bench.py --mode partitioned [--force-exchange --lanes L]` is the full form with the collectives."""
    from tch_geometric import partitioned
    shard = partitioned.CscShard.from_full(ptrs, indices, 0, 1)
    res = {}
    for name, slots in (("slot_replies", None), ("compact_replies", False)):
        ps = partitioned.PartitionedSampler(shard, G, B, fanout, slot_replies=slots)
        seeds = _cabi.seed_batches(0xBA7C4, first, G, B, n_nodes, dev)
        for _ in range(2):
            out = ps.sample(seeds, 0, first)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = ps.sample(seeds, 0, first)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        edges = int(out.counts[:, 1].sum())
        slots_n = int(out.layer_offsets[:, len(fanout) - 1, 0].sum()) if fanout else 0
        entry = {"ms_per_call": ms, "sampled_edges": edges, "edges_per_s": edges / ms * 1e3, "uses_slot_replies": bool(ps.slots),
                 "roofline": roofline_block(24 * slots_n + 40 * edges + 16 * B * G, ms,
                                            "the replicated operator's bytes (SURVEY 8d): requests and replies are the "
                                            "protocol's own traffic and count as waste")}
        if name == "slot_replies":   # same results as the replicated launch
            ref = _cabi.NsBatchedOut(G, B, fanout, dev)
            _cabi.ns_homo_batched(_cabi.graph_view(ptrs, indices), seeds, fanout, 0, first, ref)
            torch.cuda.synchronize()
            same = bool(torch.equal(out.counts, ref.counts)) and bool(torch.equal(out.layer_offsets, ref.layer_offsets))
            ar = torch.arange(out.edge_index.shape[1], device=dev)[None, :]
            same = same and bool(((out.edge_index == ref.edge_index) | (ar >= ref.counts[:, 1:2])).all())
            entry["equals_replicated_launch"] = same
            if not same:
                raise RuntimeError("mode 2 differs from the replicated launch")
            del ref
        res[name] = entry
        del ps, out
        torch.cuda.empty_cache()
    return res


def roofline_block(alg_bytes, ms, what):
    """`roofline` of a secondary configuration: algorithmic bytes / HIP-event time against the HBM peak."""
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes": alg_bytes, "ms": ms, "bytes_rule": what}


def launch_size_sweep(torch, _cabi, graph, dev, B, fanout, n_nodes, first, sizes=(256, 1024, 4096)):
    """tg_ns_homo_batched_ws at the launch sizes a loader uses: per size the form AUTO takes and, beside it, the fused
    per-batch kernel and the window-ordered form forced -- sampled edges/s and roofline fraction each (HIP events)."""
    res = {}
    base = _cabi.ns_win_tuning()
    _cabi.ns_win_tuning_set(staged=2, stage_parts=1)   # "auto" = the library's own defaults, not the headline's pick
    defaults = _cabi.ns_win_tuning()
    for G in sizes:
        out = _cabi.NsBatchedOut(G, B, fanout, dev)
        ws = _cabi.ns_homo_workspace(G, B, fanout, dev, staged=True, graph=graph)
        seeds = _cabi.seed_batches(0xBA7C4, first, G, B, n_nodes, dev)
        _cabi.ns_homo_batched(graph, seeds, fanout, 0, first, out, form=2)
        torch.cuda.synchronize()
        edges = int(out.counts[:, 1].sum())
        slots = int(out.layer_offsets[:, len(fanout) - 1, 0].sum()) if fanout else 0
        alg = 24 * slots + 40 * edges + 16 * B * G
        entry = {"sampled_edges": edges, "algorithmic_bytes": alg}
        for name, form, knobs in (("auto", 0, {}), ("fused", 2, {}), ("windowed_push", 1, dict(staged=0)),
                                  ("windowed_staged", 1, dict(staged=1, stage_parts=1))):
            _cabi.ns_win_tuning_set(**knobs)
            try:
                reps = max(4, 8192 // G)
                best = None
                for _ in range(2):
                    for _w in range(2):
                        _cabi.ns_homo_batched(graph, seeds, fanout, 0, first, out, ws=ws, form=form)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _r in range(reps):
                        _cabi.ns_homo_batched(graph, seeds, fanout, 0, first, out, ws=ws, form=form)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / reps
                    best = ms if best is None else min(best, ms)
                taken = _cabi.ns_homo_batched_form(graph, out, G, B, fanout, ws=ws, form=form)[0]
                staged = bool(_cabi.ns_homo_batched_staged(graph, out, G, B, fanout, ws=ws, form=form))
            finally:
                _cabi.ns_win_tuning_set(**defaults)
            entry[name] = {"ms_per_launch": best, "edges_per_s": edges / best * 1e3,
                           "roofline_frac": alg / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "takes": "fused" if taken == 2 else ("windowed, staged" if staged else "windowed, push")}
        res["%d_batches" % G] = entry
        del out, ws, seeds
        torch.cuda.empty_cache()
    _cabi.ns_win_tuning_set(**base)
    return res


def gpu_state(index):
    """sclk / mclk / power of this rank's GPU from sysfs (so that box-to-box spread can be attributed); never fails."""
    st = {}
    try:
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        if not cards:
            return {"unavailable": "no amdgpu sysfs"}
        base = os.path.dirname(cards[min(index, len(cards) - 1)])

        def active(name):
            for line in open(os.path.join(base, name)).read().splitlines():
                if line.rstrip().endswith("*"):
                    return line.split(":", 1)[1].strip().rstrip("*").strip()
            return None

        st["sclk"], st["mclk"] = active("pp_dpm_sclk"), active("pp_dpm_mclk")
        for hw in glob.glob(os.path.join(base, "hwmon", "hwmon*")):
            for key, fn in (("power_w", "power1_average"), ("power_w", "power1_input"), ("power_cap_w", "power1_cap")):
                f = os.path.join(hw, fn)
                if os.path.exists(f) and key not in st:
                    try:
                        st[key] = int(open(f).read()) / 1e6
                    except Exception:
                        pass
    except Exception as e:  # noqa: BLE001
        st["unavailable"] = repr(e)
    return st


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def secondary_configs(torch, _cabi, dev, cpu_seconds):
    """Short runs of the other BASELINE.json configurations on this box (reported beside the headline, never part of
    `value`): cfg3 random_walk 1 M x 80 on RMAT-24, cfg4 heterogeneous sampling + hgt_sampling (3 node types / 5
    relations) -- each with the CPU port (oracle ref-mode, one thread, bounded sample) timed beside it."""
    sec = {}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc  # the checker, used here only as the timed CPU baseline

    def timed(fn, reps=3):
        fn(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            o = fn(r + 1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, o

    budget = max(1.0, cpu_seconds / 3)
    try:  # cfg3
        n = 1 << 24
        row, col = _cabi.rmat_edges(24, n * 16, 0x5EED0000 + 24, dev)
        ptrs, idx, _ = _cabi.coo_to_csx(row, col, n, n, False)
        del row, col
        start = _cabi.seed_batches(0x57A27, 0, 1, 1 << 20, n, dev)[0].contiguous()
        g = _cabi.graph_view(ptrs, idx)
        ms, w = timed(lambda c: _cabi.random_walk(g, start, 80, 1.0, 1.0, 0, c))
        steps = int((w[:, 1:] >= 0).sum().item())
        entry = {"ms": ms, "executed_steps": steps, "steps_per_s": steps / ms * 1e3}
        n_w, L = int(w.shape[0]), int(w.shape[1]) - 1
        entry["roofline"] = roofline_block(32 * steps + 8 * n_w + 8 * (n_w * L - steps), ms,
                                           "32 B per executed step (16 ptrs + 8 neighbour + 8 written) + 8 B per walker (start "
                                           "read) + 8 B per unexecuted cell (the -1 fill)")
        hp, hi, hs = ptrs.cpu().numpy(), idx.cpu().numpy(), start.cpu().numpy()
        nw, t0 = 4096, time.perf_counter()
        ref = orc.random_walk(hp, hi, hs[:nw], 80, 1.0, 1.0, orc.rng_ref_child(orc.rng_ref()))
        dt = time.perf_counter() - t0
        nw2 = int(min(len(hs), max(nw, nw * budget / max(dt, 1e-4))))
        t0 = time.perf_counter()
        ref = orc.random_walk(hp, hi, hs[:nw2], 80, 1.0, 1.0, orc.rng_ref_child(orc.rng_ref()))
        dt = time.perf_counter() - t0
        cs = int((ref[:, 1:] >= 0).sum())
        entry["cpu_baseline"] = {"value": cs / dt, "unit": "steps/s", "cores": 1, "kind": "port",
                                 "sample": "first %d of the 1 M walkers, oracle ref-mode, %.1f s" % (nw2, dt)}
        sec["cfg3_random_walk_1M_x_80_p1_q1"] = entry
        del ptrs, idx, w, g, hp, hi
    except Exception as e:  # noqa: BLE001  (a secondary run must never cost the headline line)
        sec["cfg3_random_walk_1M_x_80_p1_q1"] = {"error": repr(e)}
    try:  # cfg4
        torch.cuda.empty_cache()
        scales = {"A": 23, "B": 22, "C": 22}
        node_types = ["A", "B", "C"]
        edge_types = [("A", "e0", "A"), ("A", "e1", "B"), ("B", "e2", "A"), ("B", "e3", "C"), ("C", "e4", "A")]
        tix = {"A": 0, "B": 1, "C": 2}
        rels, P, I = [], {}, {}
        for r, (s_, _, d_) in enumerate(edge_types):
            rw, cl = _cabi.rmat_edges_rect(scales[s_], scales[d_], 20_000_000, 0xC0F4 + r, dev)
            p_, i_, _ = _cabi.coo_to_csx(rw, cl, 1 << scales[s_], 1 << scales[d_], True)
            rels.append((tix[s_], tix[d_], p_, i_, [15, 10]))
            P["%s__%s__%s" % edge_types[r]], I["%s__%s__%s" % edge_types[r]] = p_, i_
        nb = 512
        sd = _cabi.seed_batches(0xBA7C4, 5000, nb, 1024, 1 << 23, dev)
        hb = _cabi.NsHeteroBatched(3, rels, [sd, None, None], 2, nb, dev)
        ms, _ = timed(lambda c: hb.run(0, c * nb))
        ne = int(hb.counts[:, 3:].sum().item())
        entry = {"ms_per_launch": ms, "sampled_edges": ne, "edges_per_s": ne / ms * 1e3}
        # frontier slots of relation r in hop h = samples of its dst type that arrived during the previous hop (the seeds in
        # hop 0): layer_offsets[b, r, h] = (len(samples[src]), len(edges[r]), len(samples[dst])) when hop h starts
        lo = hb.layer_offsets  # [nb, R, H, 3]
        dst_before = torch.cat([torch.zeros_like(lo[:, :, :1, 2]), lo[:, :, :-1, 2]], dim=2)
        slots = int((lo[:, :, :, 2] - dst_before).sum().item())
        entry["roofline"] = roofline_block(24 * slots + 40 * ne + 16 * nb * 1024, ms,
                                           "per relation and hop 24 B per frontier slot + 40 B per sampled edge; 16 B per seed")
        hP = {k: v.cpu().numpy() for k, v in P.items()}
        hI = {k: v.cpu().numpy() for k, v in I.items()}
        hs = sd.cpu().numpy()
        nn = {k: [15, 10] for k in hP}
        t0, done, ce = time.perf_counter(), 0, 0
        parent = orc.rng_ref()
        while done < nb and time.perf_counter() - t0 < budget:
            o = orc.ns_hetero(node_types, edge_types, hP, hI, {"A": hs[done]}, nn, 2, orc.rng_ref_child(parent))
            ce += sum(len(v) for v in o[1].values())
            done += 1
        dt = time.perf_counter() - t0
        entry["cpu_baseline"] = {"value": ce / dt, "unit": "edges/s", "cores": 1, "kind": "port",
                                 "sample": "%d of the 512 mini-batches, oracle ref-mode, %.1f s" % (done, dt)}
        sec["cfg4_neighbor_sampling_heterogenous_512_batches"] = entry
        # hgt_sampling through the operator surface, as a DataLoader worker calls it
        import tch_geometric as tg
        seeds = sd[0].contiguous()
        ns = {t: [512, 512] for t in node_types}
        tg.seed(1)

        def hgt_call(c):
            return tg.hgt_sampling(node_types, edge_types, P, I, None, {"A": seeds}, None, ns, 2)

        ms, out = timed(hgt_call, reps=5)
        nodes = sum(int(v.numel()) for v in out[0].values())
        edges = sum(int(v.numel()) for v in out[2].values())
        entry = {"ms_per_call": ms, "sampled_nodes": nodes, "sampled_edges": edges,
                 "nodes_plus_edges_per_s": (nodes + edges) / ms * 1e3}
        # per budget update of a node, per relation INTO its type: 16 B (ptrs) + 8 B x min(deg, 50); the edge
        # rebuild reads the same per output node again (hash-map traffic excluded).  Every sampled node is updated once
        # except those of the last layer; every output node is rebuilt.
        hgt_bytes = 0
        n_last = {t: ns[t][-1] for t in node_types}
        for (s_, r_, d_) in edge_types:
            p_ = P["%s__%s__%s" % (s_, r_, d_)]
            w_ = out[0][d_]
            deg = (p_[w_ + 1] - p_[w_]).clamp(max=50)
            upd = w_.numel() - min(n_last[d_], max(w_.numel() - int(seeds.numel() if d_ == "A" else 0), 0))
            hgt_bytes += int((16 * upd + 8 * deg[:upd].sum()).item()) + int((16 * w_.numel() + 8 * deg.sum()).item())
        entry["roofline"] = roofline_block(hgt_bytes, ms, "16 B + 8 B x min(deg, 50) per (node, relation into its type) per budget "
                                           "update and again per output node for the edge rebuild; hash maps excluded -- the "
                                           "call is bound by its chain of ~48 small launches, not by bytes")
        t0, done = time.perf_counter(), 0
        while done < 64 and time.perf_counter() - t0 < budget:
            orc.hgt(node_types, edge_types, hP, hI, None, {"A": hs[done]}, None, ns, 2, orc.rng_ref_child(parent))
            done += 1
        dt = time.perf_counter() - t0
        entry["cpu_baseline"] = {"value": done / dt, "unit": "calls/s", "gpu_calls_per_s": 1e3 / ms, "cores": 1,
                                 "kind": "port", "sample": "%d calls, oracle ref-mode, %.1f s" % (done, dt)}
        sec["cfg4_hgt_sampling_1024_seeds_512x2_per_type"] = entry
    except Exception as e:  # noqa: BLE001
        sec.setdefault("cfg4_neighbor_sampling_heterogenous_512_batches", {"error": repr(e)})
        sec.setdefault("cfg4_hgt_sampling_1024_seeds_512x2_per_type", {"error": repr(e)})
    return sec


def cpu_baseline(args, ptrs, indices, seeds, fanout):
    """The oracle's ref-mode (sequential Xoshiro256++, the reference's algorithm draw for draw) timed on
    this box's host cores over a bounded sample of the same batches; threads own whole batches (how PyG's
    `num_workers` would scale the reference).  ALL host cores are used; the one-core figure is kept beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc  # the checker, used here only as the timed CPU baseline

    hp, hi = ptrs.cpu().numpy(), indices.cpu().numpy()
    hs = seeds.cpu().numpy()
    nproc = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = nproc
    try:  # a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) bounds the cores that really run
        quota = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else int(q) / int(per)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        if quota:
            usable = max(1, min(usable, int(round(quota))))
    except Exception:
        pass
    threads = max(1, usable)
    # Two builds of the same source are timed and the FASTER one is the stated baseline (the denominator must
    # not drift with a build flag): the checker that travels with the repo (-O3 -march=x86-64-v2; it must run on whatever
    # CPU the GPU box has) and the same file compiled here, on this host, with -O3 -march=native.
    builds = [("portable", None, "gcc (oracle/Makefile build)", orc.PORTABLE_CFLAGS)]
    nat = orc.native_lib()
    if nat:
        builds.append(("native", nat[0], nat[1], nat[2]))
    per_build = {}
    share = args.cpu_seconds / len(builds)
    for name, handle, cc, cflags in builds:
        n0 = min(threads, hs.shape[0])
        sec0, _ = orc.bench_ns_homo(hp, hi, hs[:n0], fanout, threads, handle)  # calibration: one batch per thread
        per_round = max(sec0, 1e-4)
        n = int(min(hs.shape[0], max(n0, threads * max(1, round(share / per_round)))))
        sec, edges = orc.bench_ns_homo(hp, hi, hs[:n], fanout, threads, handle)
        n1 = max(1, min(n // threads, 64))
        sec1, edges1 = orc.bench_ns_homo(hp, hi, hs[:n1], fanout, 1, handle)
        per_build[name] = {"cc": cc, "cflags": cflags, "value": edges / sec, "single_thread_value": edges1 / sec1,
                           "batches": n, "wall_s": round(sec, 2), "single_thread_batches": n1}
    best = max(per_build, key=lambda k: per_build[k]["value"])
    b = per_build[best]
    return {
        "cc": b["cc"],
        "cflags": b["cflags"],
        "value": b["value"],
        "unit": "edges/s",
        "cores": threads,
        "nproc": nproc,
        "cpu_model": cpu_model(),
        "kind": "port",
        "build": best,
        "builds": per_build,
        "sample": "%d of the 1024-seed mini-batches of the timed steps, oracle ref-mode (rand-0.8.5 Xoshiro256++ stream, "
                  "reservoir loop of sampling.rs), %d threads (every core this process may use: affinity and cgroup quota; nproc %d) each owning "
                  "whole batches, %.1f s wall; single thread: %.3g edges/s over %d batches; the faster of the two builds timed (%s)" %
                  (b["batches"], threads, nproc, b["wall_s"], b["single_thread_value"], b["single_thread_batches"],
                   ", ".join("%s %.3g" % (k, v["value"]) for k, v in per_build.items())),
        "single_thread_value": max(v["single_thread_value"] for v in per_build.values()),
    }


if __name__ == "__main__":
    main()
