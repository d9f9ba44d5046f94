"""Headline benchmark: DNA tokens/s of one hg38-style MLM pre-training step (forward + backward + gradient all-reduce +
clip + AdamW) of Caduceus-PS d_model=256 n_layer=16 at seqlen 131072, bf16 compute, one sequence per GPU
(BASELINE.json configs[2] at N=1, configs[3] at N=8; /root/reference/slurm_scripts/run_pretrain_caduceus.sh:19-60).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel (the selective scan) from HIP-event timings
taken inside the timed region; `cpu_baseline` times the CPU oracle (C/OpenMP scan + torch CPU GEMMs) on a bounded
sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

COMP = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 6, 7: 10, 8: 9, 9: 8, 10: 7, 11: 11}  # tokenization_caduceus.py:49-66
SSM_CFG = dict(d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=1e-3, dt_max=0.1, dt_init="random",
               dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guide: 8.0 TB/s; 6.29 TB/s measured streaming copy)


def make_config(d_model, n_layer, rcps=True):
    from caduceus_amd import CaduceusConfig
    return CaduceusConfig(d_model=d_model, n_layer=n_layer, vocab_size=12, ssm_cfg=dict(SSM_CFG), rms_norm=True,
                          residual_in_fp32=False, fused_add_norm=True, pad_vocab_size_multiple=8, norm_epsilon=1e-5,
                          initializer_cfg=dict(initializer_range=0.02, rescale_prenorm_residual=True,
                                               n_residuals_per_layer=1),
                          bidirectional=True, bidirectional_strategy="add", bidirectional_weight_tie=True, rcps=rcps,
                          complement_map=dict(COMP), pad_token_id=4)


def synthetic_batch(gen, B, L, device):
    """uniform A/C/G/T (ids 7..10), 0.1% N -> pad(4); MLM corruption of
    src/dataloaders/utils/mlm.py:4-32 (15% targets: 80% [MASK]=3, 10% random id, 10% unchanged; other labels = 4)."""
    ids = torch.randint(7, 11, (B, L), generator=gen)
    ids[torch.rand(B, L, generator=gen) < 0.001] = 4
    labels = ids.clone()
    tgt = torch.rand(B, L, generator=gen) < 0.15
    labels[~tgt] = 4
    r = torch.rand(B, L, generator=gen)
    ids = ids.clone()
    ids[tgt & (r < 0.8)] = 3
    rnd = tgt & (r >= 0.8) & (r < 0.9)
    ids[rnd] = torch.randint(0, 12, (int(rnd.sum()),), generator=gen)
    return ids.to(device), labels.to(device)


def host_topology():
    """Physical cores of socket 0 (one hardware thread each) and the CPU model string, from sysfs / procfs.  The CPU baseline is pinned
    to exactly these: with every hardware thread of a two-socket host in one OpenMP team the 1024-token problem of configs[0] measured
    thread synchronisation across sockets, and swung 4x from box to box."""
    cpus, seen = [], set()
    allowed = sorted(os.sched_getaffinity(0))
    first_pkg = min_pkg(allowed)  # (once: one sysfs read per CPU, not per pair of CPUs)
    for c in allowed:
        base = f"/sys/devices/system/cpu/cpu{c}/topology/"
        try:
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
        except (OSError, ValueError):
            pkg, core = 0, c
        if pkg == first_pkg and (pkg, core) not in seen:
            seen.add((pkg, core))
            cpus.append(c)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cpus or allowed, model


def min_pkg(allowed):
    pk = []
    for c in allowed:
        try:
            pk.append(int(open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id").read()))
        except (OSError, ValueError):
            pk.append(0)
    return min(pk) if pk else 0


def cpu_baseline(scan_L=131072):
    """Runs cpu_baseline_worker in a fresh process pinned to the physical cores of one socket (OMP_NUM_THREADS = that count,
    OMP_PROC_BIND=close, OMP_PLACES=cores, sched_setaffinity before any OpenMP runtime starts) and returns its JSON."""
    import subprocess
    cpus, model = host_topology()
    env = dict(os.environ, OMP_NUM_THREADS=str(len(cpus)), OMP_PROC_BIND="close", OMP_PLACES="cores", MKL_NUM_THREADS=str(len(cpus)),
               CAD_CPU_BASELINE_CPUS=",".join(map(str, cpus)), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(int(scan_L))], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, check=False)
    if out.returncode != 0:
        raise RuntimeError("cpu baseline worker failed: " + out.stderr.decode()[-400:])
    res = json.loads(out.stdout.decode().strip().splitlines()[-1])
    res["cpu_model"] = model
    return res


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def cpu_baseline_worker(scan_L=131072, reps_model=15, reps_scan=3):
    """the oracle (reference formulation, fp32, C/OpenMP scan + torch CPU GEMMs) on the host cores:
    (i) configs[0] = Caduceus-PS d_model=128 n_layer=4 seqlen=1024, batch 1, one full forward + backward -> tokens/s
    (`value` = 1024 / the MEDIAN of `reps_model` timed repetitions after two warm-up passes); (ii) the selective-scan op alone at
    (B, E, L, N) = (1, 512, scan_L, 16), forward and backward (median of `reps_scan`) -> GB/s of the same algorithmic bytes the GPU
    roofline uses (fp32 on the CPU: s = 4)."""
    cpus = [int(c) for c in os.environ.get("CAD_CPU_BASELINE_CPUS", "").split(",") if c]
    if cpus:
        os.sched_setaffinity(0, cpus)  # before torch / the oracle library start their OpenMP teams
    import torch as th
    if cpus:
        th.set_num_threads(len(cpus))
    from caduceus_amd import CaduceusForMaskedLM
    from oracle import oracle_model as om
    from oracle import oracle_ops
    th.manual_seed(2222)
    d_model, n_layer, L = 128, 4, 1024
    cfgobj = make_config(d_model, n_layer)
    model = CaduceusForMaskedLM(cfgobj)  # parameter container only (CPU); arithmetic below is the oracle's
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v)
          for k, v in model.state_dict().items()}
    cfg = dict(rcps=True, fused_add_norm=True, rms_norm=True, norm_epsilon=1e-5, n_layer=n_layer, bidirectional=True,
               bidirectional_strategy="add")
    ids, labels = synthetic_batch(th.Generator().manual_seed(1), 1, L, "cpu")
    om.set_scan_backend(oracle_ops.selective_scan_c)
    times = []
    try:
        for rep in range(2 + reps_model):  # two warm-up passes (thread pools, allocator), then the timed ones
            t0 = time.perf_counter()
            out = om.masked_lm_forward(sd, ids, cfg, labels=labels, ignore_index=4)
            out["loss"].backward()
            if rep >= 2:
                times.append(time.perf_counter() - t0)
    finally:
        om.set_scan_backend(None)
    c1 = L / _median(times)
    # scan-op microbench (SURVEY 8d inputs: u, z, delta_raw, B, C ~ N(0,1); A = -(1..16); D = 1)
    E, N = 512, 16
    g = th.Generator().manual_seed(3)
    r = lambda *sh: th.randn(*sh, generator=g)  # noqa: E731
    u, delta, z, Bm, Cm = r(1, E, scan_L), r(1, E, scan_L), r(1, E, scan_L), r(1, N, scan_L), r(1, N, scan_L)
    A = -(th.arange(1, N + 1).float().repeat(E, 1))
    D, bias = th.ones(E), r(E) - 4.0
    ins = [t.requires_grad_(True) for t in (u, delta, A, Bm, Cm, D, z, bias)]
    oracle_ops.selective_scan_c(*ins)  # warm-up
    tfs, tbs = [], []
    for _ in range(reps_scan):
        t0 = time.perf_counter()
        y = oracle_ops.selective_scan_c(*ins)
        tfs.append(time.perf_counter() - t0)
        go = th.randn(y.shape, generator=g)
        t0 = time.perf_counter()
        y.backward(go)
        tbs.append(time.perf_counter() - t0)
    tf, tb = _median(tfs), _median(tbs)
    threads = max(oracle_ops.num_threads(), th.get_num_threads())
    res = {"value": c1, "unit": "tokens/s", "cores": threads, "kind": "port",
           "sample": f"configs[0] Caduceus-PS d_model=128 n_layer=4 seqlen=1024 batch=1, full fwd+bwd: median of {len(times)} "
                     f"repetitions after 2 warm-up passes (fp32, oracle/oracle_model.py + oracle/cad_oracle.c OpenMP scan), "
                     f"{sum(times):.1f} s timed; pinned to the {len(cpus) or threads} physical cores of one socket "
                     f"(OMP_PROC_BIND=close, OMP_PLACES=cores); os.cpu_count()={os.cpu_count()}",
           "rep_seconds": [round(t, 4) for t in times], "spread": (max(times) - min(times)) / _median(times),
           "scan_microbench": {"shape_BELN": [1, E, scan_L, N], "dtype": "f32", "fwd_s": tf, "bwd_s": tb, "reps": reps_scan,
                               "fwd_GBps": (4 * E + 2 * N) * 4 * scan_L / tf / 1e9,
                               "bwd_GBps": (7 * E + 4 * N) * 4 * scan_L / tb / 1e9}}
    print(json.dumps(res), flush=True)


# Issue price of one wave instruction on one SIMD with two resident waves, ns, by instruction class -- this chip's micro-benchmark
# (tools/ubench/ubench.hip, profiles/r01_ubench_gfx950.log: SIMD ticks per instruction at waves/SIMD = 2: v_mul / v_mov 2.2, v_fma 2.8,
# v_pk_* and v_cvt_pk 3.6, VOP2 + DPP 3.5, v_exp / v_log / v_rcp 6.2, at the ~1.85 ticks per ns of that run; re-measured in round 6 on the
# final build's box, profiles/r06_ubench_gfx950.log: the same ns per instruction to two digits).  "plain" = non-packed
# VALU incl. moves, selects and v_readlane (between v_mul and v_fma).
VALU_PRICE_NS = {"valu": 1.15, "valu_mov": 1.0, "valu_sel": 1.0, "valu_lane": 1.0, "valu_pk": 1.95, "valu_dpp": 1.9, "valu_cvt": 1.98,
                 "trans": 3.4}
# static instruction mix of the production scan kernels (tools/make_scan_isa_json.py), quoted like the counter profile: only for the
# scan sources it was counted on
SCAN_ISA_FILE = os.path.join("profiles", "r06_scan_isa.json")


def valu_roofline(kind: str, pmc: dict, isa: dict, avg_ms: float, n_simds: int):
    """The ceiling that actually binds the scans: VALU issue.  Executed VALU wave-instructions per launch
    (rocprofv3 SQ_INSTS_VALU, transcendentals included) priced with the class mix of the kernel's chunk loop (static count) and this
    chip's issue prices, spread over the SIMDs; next to it how busy the counters say the VALU was."""
    sq = (pmc or {}).get("sq", {}).get(kind)
    mix = (isa or {}).get("kernels", {}).get(kind, {}).get("chunk_loop_static")
    if not sq or not mix:
        return None
    cls = {k: v for k, v in mix.items() if k in VALU_PRICE_NS}
    n_static = sum(cls.values())
    mean_price = sum(VALU_PRICE_NS[k] * v for k, v in cls.items()) / n_static  # ns per executed VALU wave-instruction, by the static mix
    insts = sq["SQ_INSTS_VALU"]
    ceiling_ms = insts * mean_price / n_simds * 1e-6
    return {"insts_valu_per_launch": insts, "insts_valu_per_wave": insts / max(1.0, sq.get("SQ_WAVES", 0.0)),
            "static_chunk_loop_mix": mix, "positions_per_chunk": isa["kernels"][kind]["positions_per_chunk"],
            "vgprs": isa["kernels"][kind]["vgprs"], "scratch_bytes": isa["kernels"][kind]["scratch_bytes"],
            "price_ns_per_wave_instruction": VALU_PRICE_NS, "mean_price_ns": mean_price, "simds": n_simds,
            "issue_ceiling_ms": ceiling_ms, "kernel_over_issue_ceiling": avg_ms / ceiling_ms,
            # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: share of a resident wave's lifetime in which one of ITS VALU instructions executes
            # (two waves per SIMD: twice this value is the SIMD's VALU-busy share); SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: parked at s_waitcnt
            "valu_active_share_of_wave_cycles": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"] if sq.get("SQ_WAVE_CYCLES") else None,
            "waitcnt_share_of_wave_cycles": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"] if sq.get("SQ_WAVE_CYCLES") else None,
            "lds_bank_conflict_share": (sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]) if sq.get("SQ_LDS_IDX_ACTIVE") else None}


# counter profile of the two scan kernels at the headline launch shape (tools/prof_scan.sh + tools/make_scan_pmc_json.py), stamped with
# the library and scan-source hashes it was taken on; quoted in `roofline` only while pmc_quotable() holds
SCAN_PMC_FILE = os.path.join("profiles", "r06_scan_pmc.json")


def pmc_quotable(pmc: dict, lib_version: str):
    """A committed counter profile of the scans may be quoted next to a timing of THIS library when it was taken on this very build,
    or on a build whose SCAN sources (scan_*.hip, scan_common.h, cad_common.h, the C-ABI header: `_build.scan_source_hash`) are the
    ones the loaded library was built from (i.e. the library is the build of the tree the hash is computed on).  Returns the
    provenance text, or None."""
    from caduceus_amd import _build
    if pmc.get("lib_version") == lib_version:
        return "this build (" + lib_version + ")"
    if pmc.get("scan_src") and pmc.get("scan_src") == _build.scan_source_hash() and lib_version.endswith(_build.source_hash()):
        return ("build " + str(pmc.get("lib_version")) + " with the same scan sources (scan_src " + str(pmc.get("scan_src")) +
                ") as this library, " + lib_version)
    return None


def scan_floor_worker(d_model, seqlen, strands, batch, reps=4):
    """Runs in a child process whose CADUCEUS_AMD_LIB is the arithmetic-only timing build (caduceus_amd/_build.py::build_floor): ONE
    production mixer layer (forward + backward, exactly as the training step launches it) at the benchmark's layer shape, the two scan
    kinds timed by the library's HIP events.  Prints {"scan_fwd_ms": .., "scan_bwd_ms": ..} per scan OPERATION of a layer."""
    from caduceus_amd import _lib, mixer
    from caduceus_amd.mamba import Mamba
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mf, mr = Mamba(d_model, device=dev), Mamba(d_model, device=dev)
    mr.in_proj.weight = mf.in_proj.weight
    mr.out_proj.weight = mf.out_proj.weight
    hn = torch.randn(strands, batch, seqlen, d_model, device=dev).to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(strands, batch, seqlen, d_model, device=dev).to(torch.bfloat16)
    split = batch if strands == 2 else strands * batch

    def step():
        for p in list(mf.parameters()) + list(mr.parameters()):
            p.grad = None
        hn.grad = None
        mixer.prepare_step_cache([(mf, mr)], torch.bfloat16)
        mixer.bimamba_mixer(hn, mf, mr, split).backward(g)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True, kinds=("scan_fwd", "scan_bwd"))
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    pr = _lib.prof_read()
    print(json.dumps({"lib": _lib.version(), "scan_fwd_ms": pr["scan_fwd"][0] / reps, "scan_bwd_ms": pr["scan_bwd"][0] / reps}), flush=True)


def scan_floor(d_model, seqlen, strands, batch):
    """{"scan_fwd_ms", "scan_bwd_ms"} of the arithmetic-only build on this GPU, or None when that library has not been built."""
    import subprocess
    from caduceus_amd import _build
    if not os.path.exists(_build.FLOOR_LIB):
        return None
    env = dict(os.environ, CADUCEUS_AMD_LIB=_build.FLOOR_LIB, CADUCEUS_AMD_ALLOW_TIMING_BUILD="1")  # (caduceus_amd/_lib.py refuses it otherwise)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--scan-floor-worker", str(d_model), str(seqlen), str(strands),
                          str(batch)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, check=False)
    if out.returncode != 0:
        return {"error": out.stderr.decode()[-300:]}
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--cpu-baseline-worker":
        cpu_baseline_worker(int(sys.argv[2]) if len(sys.argv) > 2 else 131072)
        return
    if len(sys.argv) >= 6 and sys.argv[1] == "--scan-floor-worker":
        scan_floor_worker(*[int(v) for v in sys.argv[2:6]])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seqlen", type=int, default=131072)
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--n-layer", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1, help="sequences per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cpu-sample", type=int, default=131072,
                    help="length of the CPU scan microbench (SURVEY 8d: 131072); 0 = skip the CPU baseline")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="sequences per OPTIMIZER step over all GPUs (reference: 8, configs/experiment/hg38/hg38.yaml:17 "
                         "accumulate_grad_batches = global / (gpus * batch)); 0 = weak scaling, one micro-batch per step")
    ap.add_argument("--fp8-proj", action="store_true",
                    help="configs[4]: in_proj on the fp8 (OCP e4m3) matrix cores -- per-token activation scales, per-row weight "
                         "scales, fp32 accumulation (csrc/gemm_fp8.hip); everything else stays bf16")
    ap.add_argument("--no-floor", action="store_true", help="skip the arithmetic-only scan timing (roofline.arithmetic_floor)")
    ap.add_argument("--model", default="ps", choices=["ps", "ph"],
                    help="ps: RCPS (configs[2..4], the headline); ph: no RCPS wrapper, RC augmentation is a data-side flip "
                         "(configs[1], run with --seqlen 1024 --batch 128)")
    args = ap.parse_args()

    if args.fp8_proj:
        if args.dtype != "bf16":
            raise SystemExit("--fp8-proj runs inside the bf16 path")
        os.environ["CADUCEUS_AMD_FP8_PROJ"] = "1"
    import torch.distributed as dist
    from caduceus_amd import CaduceusForMaskedLM, _lib
    from caduceus_amd.dp import BucketedGradReducer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    # TEST HOOK (tests/test_bench_cli.py, CPU suite): CADUCEUS_BENCH_TEST_BACKEND=emu runs this very script -- rank / world handling,
    # per-rank seeds, accumulation under no_sync(), the MAX all-reduce of the timing, the JSON line -- on CPU tensors with the kernels
    # of the host emulator (tests/emu) and gloo, so that the N > 1 code path of bench.py has been executed before the driver
    # launches it on an 8-GPU node.  Never a measurement: the line says data = "synthetic (host emulator: not a measurement)".
    emu = os.environ.get("CADUCEUS_BENCH_TEST_BACKEND") == "emu"
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        from build_emu import build_emu
        _lib.use_library_for_testing(build_emu())
        dev = torch.device("cpu")
        args.cpu_sample = 0
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("CADUCEUS_DP_FORCE_COLLECTIVE") == "1"
    if use_dist:
        if emu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm

    torch.manual_seed(2222)  # configs/experiment/hg38/hg38.yaml:54; identical initial weights on every rank
    model = CaduceusForMaskedLM(make_config(args.d_model, args.n_layer, rcps=args.model == "ps")).to(dev).train()
    n_params = sum(p.numel() for p in model.parameters())
    reducer = BucketedGradReducer(model.parameters())
    decay, no_decay = [], []
    for n_, p in model.named_parameters():  # src/utils/optim_groups.py:25-38 semantics
        (no_decay if (n_.endswith("bias") or getattr(p, "_no_weight_decay", False) or "embedding" in n_) else decay
         ).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.1}, {"params": no_decay, "weight_decay": 0.0}],
                            lr=8e-3, betas=(0.9, 0.95), fused=True)  # (torch's fused AdamW also has a CPU implementation)
    amp = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    gen = torch.Generator().manual_seed(2222 + rank)
    batches = [synthetic_batch(gen, args.batch, args.seqlen, dev) for _ in range(min(8, args.steps + args.warmup))]

    # gradient accumulation: the reference keeps the global batch at every GPU count (hg38.yaml:17); micro-steps before
    # the last one run under no_sync() -- one all-reduce per optimizer step, overlapped with the last backward
    accum = 1
    if args.global_batch:
        if args.global_batch % (world * args.batch):
            raise SystemExit("--global-batch must be a multiple of gpus * batch")
        accum = args.global_batch // (world * args.batch)

    def micro(i, scale):
        ids, labels = batches[i % len(batches)]
        with torch.autocast(dev.type, dtype=amp, enabled=amp != torch.float32):
            out = model(ids, labels=labels)
        (out.loss * scale if scale != 1.0 else out.loss).backward()
        return out.loss

    def step(i):
        reducer.zero_grad()
        if accum > 1:
            with reducer.no_sync():
                for k in range(accum - 1):
                    micro(i * accum + k, 1.0 / accum)
        loss = micro(i * accum + accum - 1, 1.0 / accum)
        reducer.finish()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return loss

    def fence():
        if use_dist:
            dist.barrier()
        if not emu:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        loss = step(i)
    fence()
    # HIP events around the two scan kinds only: every timed launch costs two event records on the stream (~10 us of idle queue
    # each in the kernel trace) -- with all nine kinds on, the instrumentation itself was 5 % of the step it measured
    _lib.prof_reset()
    _lib.prof_enable(True, kinds=("scan_fwd", "scan_bwd"))
    reducer.time_exposed = use_dist  # two event records per step: how long the compute stream waits for all-reduces backward did not hide
    ar0 = reducer.allreduces_launched
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    ar_timed = reducer.allreduces_launched - ar0
    reducer.time_exposed = False
    exposed = reducer.exposed_ms()
    _lib.prof_enable(False)
    prof = _lib.prof_read()
    _lib.prof_reset()
    # the other kernel families: one extra, UNTIMED step with every kind on
    _lib.prof_enable(True)
    step(args.warmup + args.steps)
    fence()
    _lib.prof_enable(False)
    prof_all = _lib.prof_read()
    _lib.prof_reset()
    t = torch.tensor([elapsed, sum(exposed) / max(1, len(exposed))], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, exposed_ms = float(t[0].item()), float(t[1].item())
    dist_info = {"world_size": dist.get_world_size() if use_dist else 1, "backend": dist.get_backend() if use_dist else None,
                 "collective_library": None, "grad_allreduce_bytes_per_step": sum(b.numel() * 4 for b in reducer.buckets),
                 "buckets": len(reducer.buckets),
                 # collectives per optimizer step, counted: = buckets (accumulation micro-steps run under no_sync())
                 "allreduces_per_step": ar_timed / max(1, args.steps),
                 # MAX over ranks of the mean time the compute stream spent inside reducer.finish(): the part of the gradient
                 # all-reduce that backward did not hide (+ the launch of late buckets); null without a process group
                 "allreduce_exposed_ms_per_step": exposed_ms if use_dist else None}
    if use_dist and not emu:
        try:
            dist_info["collective_library"] = "RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as ex:  # noqa: BLE001
            dist_info["collective_library"] = repr(ex)

    if rank == 0:
        tokens = args.batch * args.seqlen * world * args.steps * accum
        E, N = 2 * args.d_model, SSM_CFG["d_state"]
        s = 2 if args.dtype == "bf16" else 4
        # one launch = (both strands x) both parameter sets (mamba_fwd, mamba_rev) of a layer = 4 (PS) / 2 (Ph) Mamba
        # invocations per token
        inv_tokens = (4 if args.model == "ps" else 2) * args.batch * args.seqlen
        alg = {"scan_fwd": (4 * E + 2 * N) * s * inv_tokens, "scan_bwd": (7 * E + 4 * N) * s * inv_tokens}
        kinds = {}
        # one scan OPERATION per layer and micro-step; with the L-split (Caduceus-Ph at batch 1) an operation is two launches
        # (map / carry pass + full pass): the algorithmic bytes are divided by the time of BOTH
        n_ops = args.n_layer * args.steps * accum
        for k in ("scan_fwd", "scan_bwd"):
            ms, n = prof[k]
            if n and ms > 0:  # (the host emulator counts launches but has no device time)
                kinds[k] = {"launches": n, "launches_per_op": n / n_ops, "avg_ms": ms / n_ops,
                            "achieved_GBps": alg[k] / (ms / n_ops * 1e-3) / 1e9, "algorithmic_bytes_per_launch": alg[k],
                            "total_ms": ms}
        dom = max(kinds, key=lambda k: kinds[k]["total_ms"]) if kinds else None
        roofline = None
        # HBM traffic of one scan launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, collected separately with
        # rocprofv3 --pmc at exactly this launch shape); null for any other shape
        # ... and only when the profile was taken on THIS build of the kernels (cad_version() carries a hash of the sources)
        traffic, traffic_note = None, "no counter profile for this launch shape"
        pmc_ok = None  # the counter profile, when it belongs to this build AND this launch shape
        try:
            pmc = json.load(open(os.path.join(ROOT, SCAN_PMC_FILE)))
            sh = pmc["shape"]
            if dom and (sh["E"], sh["L"], sh["N"], sh["dtype"]) == (E, args.seqlen, N, args.dtype) and \
                    sh["rows"] == (2 if args.model == "ps" else 1) * args.batch:
                why = pmc_quotable(pmc, _lib.version())
                if why:
                    pmc_ok = pmc
                    traffic = pmc[dom]["fetch_bytes"] + pmc[dom]["write_bytes"]
                    traffic_note = ("HBM-side bytes per launch: rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes, " +
                                    SCAN_PMC_FILE + " taken on " + why)
                else:
                    traffic_note = (f"{SCAN_PMC_FILE} was taken on another build ({pmc.get('lib_version')}); this "
                                    f"library is {_lib.version()}: not quoted")
        except (OSError, KeyError, ValueError):
            traffic = None
        if dom:
            isa = None
            try:
                from caduceus_amd import _build
                isa = json.load(open(os.path.join(ROOT, SCAN_ISA_FILE)))
                if isa.get("scan_src") != _build.scan_source_hash() or not _lib.version().endswith(_build.source_hash()):
                    isa = None  # counted on other scan sources (or this library is not the build of this tree)
            except (OSError, ValueError):
                isa = None
            n_simds = 4 * (torch.cuda.get_device_properties(dev).multi_processor_count if not emu else 256)
            valu = {k: valu_roofline(k, pmc_ok, isa, kinds[k]["avg_ms"], n_simds) for k in kinds}
            # `achieved` / `peak` / `frac` stay what the contract asks for -- algorithmic bytes per launch over the launch time against the
            # HBM peak -- but the resource that BINDS the scans is VALU issue (two waves per SIMD; `valu`, `arithmetic_floor`), not HBM
            roofline = {"bound": "valu",  # (the dominant kernels are the scans at every configuration; `valu` carries the counters where a
                                          #  counter profile of this build and launch shape exists, null elsewhere)
                        "bound_note": "frac is the HBM fraction the contract asks for; the scans are bound by VALU issue (roofline.valu: executed "
                                      "VALU wave-instructions priced with this chip's issue rates; roofline.arithmetic_floor: the same arithmetic "
                                      "timed without memory traffic), as measured in rounds 4 and 5 (DESIGN.md section 3)",
                        "kernel": dom, "achieved": kinds[dom]["achieved_GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": kinds[dom]["achieved_GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_note": traffic_note, "valu": valu,
                        "avg_launch_ms": kinds[dom]["avg_ms"], "launches": kinds[dom]["launches"],
                        "launches_per_layer_op": kinds[dom]["launches_per_op"],
                        "algorithmic_bytes_per_launch": kinds[dom]["algorithmic_bytes_per_launch"],
                        "all": {k: {"avg_ms": v["avg_ms"], "achieved_GBps": v["achieved_GBps"],
                                    "hbm_bound_ms": alg[k] / (HBM_PEAK_GBS * 1e9) * 1e3,
                                    "share_of_step": v["total_ms"] / (elapsed * 1e3)} for k, v in kinds.items()},
                        "other_kernels_ms_per_step": {k: prof_all[k][0] for k in prof_all if k not in kinds and prof_all[k][1]},
                        "other_kernels_note": "one extra untimed step with every kernel family timed"}
        # MFMA evidence for the dense projections (north_star): the in_proj GEMM of this workload, timed stand-alone
        proj = None
        try:
            if emu:
                raise RuntimeError("host emulator: no device timing")
            Tt = (2 if args.model == "ps" else 1) * args.batch * args.seqlen
            xx = torch.randn(Tt, args.d_model, device=dev, dtype=amp)
            ww = torch.randn(2 * E, args.d_model, device=dev, dtype=amp)
            from caduceus_amd import ops as _ops
            # the kernel the product path DISPATCHES for the in_proj of this width (mixer.BiMambaMixerFn.forward): d_model > 256 ->
            # cad_gemm_stream with both operands streamed (B = W_in^T from the step cache), else the W-stationary cad_proj_wxT
            from caduceus_amd import mixer as _mixer
            own = _ops.proj_supported(xx, args.d_model)  # (csrc/gemm.hip); fp32: the fp32 matrix-core kernel (csrc/gemm_f32.hip); else hipBLASLt
            kname = "cad_proj_wxT, own bf16 MFMA kernel" if own else ("cad_gemm_f32, own fp32 MFMA kernel" if amp == torch.float32 else "hipBLASLt")
            run = (lambda: _ops.proj_wxT(ww, xx)) if own else (lambda: _ops.mm(ww, xx.t()))
            if _mixer._STREAM_PROJ_D512 and args.d_model > 256 and amp == torch.bfloat16:
                wwT = ww.t().contiguous()
                if _ops.gemm_out_t(xx, wwT) is not None:
                    run = lambda: _ops.gemm_out_t(xx, wwT)  # noqa: E731
                    kname = "cad_gemm_stream (CAD_GEMM_OUT_T_BF16, col_fastest), own bf16 MFMA kernel, both operands streamed"
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            fl = 2.0 * Tt * args.d_model * 2 * E
            by = (Tt * args.d_model + 2 * E * args.d_model + Tt * 2 * E) * (2 if args.dtype == "bf16" else 4)
            proj = {"kernel": "in_proj GEMM (" + kname + ")", "ms": ms,
                    "TFLOPs": fl / ms / 1e9, "mfma_peak_TFLOPs": 2500.0 if args.dtype == "bf16" else 157.3,
                    "GBps": by / ms / 1e6, "hbm_frac": by / ms / 1e6 / HBM_PEAK_GBS,
                    "note": "K = d_model: HBM-bound (arithmetic intensity ~200 flop/B), not MFMA-bound; per-kernel MFMA-busy and "
                            "fabric-byte counters of the whole step in profiles/r06_step_pmc_summary.txt"}
            if args.fp8_proj and _ops.fp8_proj_supported(xx, args.d_model):
                wq, sw = _ops.quant_weight_fp8(ww)
                xq, sx = _ops.quant_rows_fp8(xx)
                run8 = lambda: _ops.proj_wxT_fp8(wq, sw, xq, sx)  # noqa: E731
                runq = lambda: _ops.quant_rows_fp8(xx)  # noqa: E731
                res8 = {}
                for nm, fn in (("in_proj_fp8", run8), ("quant_rows_fp8", runq)):
                    for _ in range(3):
                        fn()
                    e0.record()
                    for _ in range(10):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    res8[nm + "_ms"] = e0.elapsed_time(e1) / 10
                by8 = Tt * args.d_model + 2 * E * args.d_model + Tt * 2 * E * 2  # e4m3 operands, bf16 output
                proj["fp8"] = {**res8, "TFLOPs": fl / res8["in_proj_fp8_ms"] / 1e9, "mfma_peak_TFLOPs": 5000.0,
                               "GBps": by8 / res8["in_proj_fp8_ms"] / 1e6,
                               "kernel": "cad_proj_wxT_fp8 (v_mfma_f32_16x16x32_fp8_fp8, per-token / per-row scales)"}
            del xx, ww
        except Exception as ex:  # evidence only
            proj = {"error": repr(ex)}
        if roofline is not None:
            roofline["projections"] = proj
            # third ceiling: the scans' arithmetic-only timing build on THIS GPU, same launch shape and power state
            floor = None
            if world == 1 and not emu and args.dtype == "bf16" and not args.no_floor:
                try:
                    floor = scan_floor(args.d_model, args.seqlen, 2 if args.model == "ps" else 1, args.batch)
                except Exception as ex:  # evidence only
                    floor = {"error": repr(ex)}
            if floor and "scan_bwd_ms" in floor:
                floor["kernel_over_floor"] = {k: kinds[k]["avg_ms"] / floor[k + "_ms"] for k in kinds if floor.get(k + "_ms")}
                floor["note"] = ("libcaduceus_hip_floor.so = the same sources with -DSC_WHATIF=14434 (SC_WHATIF_ARITH_ONLY: no global stores, no "
                                 "dB/dC slab or flush, no B/C tile loads / conversion / LDS traffic, no barrier): one production mixer layer "
                                 "in a child process right after the timed region; per scan operation of a layer")
            roofline["arithmetic_floor"] = floor
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            try:
                cpu = cpu_baseline(args.cpu_sample)
            except Exception as ex:  # the baseline is a reported number, never a reason to lose the GPU line
                cpu = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                       "sample": f"failed: {ex!r}"}
        line = {
            "metric": "DNA tokens/sec (whole node), hg38-style MLM pre-train step", "value": tokens / elapsed,
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None, "dtype": args.dtype + (" (in_proj forward: e4m3 activations [per-token scales, written by the add + norm epilogue] x e4m3 weights "
                                            "[per-row scales] on v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulate; every other operand and the "
                                            "whole backward bf16)" if args.fp8_proj else ""),
            "data": "synthetic" + (" (host emulator: not a measurement)" if emu else ""),
            "config": {"workload": f"Caduceus-{args.model.upper() if args.model == 'ps' else 'Ph'} d_model={args.d_model} "
                                   f"n_layer={args.n_layer} seqlen={args.seqlen} rcps={'true' if args.model == 'ps' else 'false'} "
                                   f"MLM fwd+bwd+allreduce+AdamW, {args.batch} seq/GPU"
                                   + (f" x {accum} accumulation micro-steps (global batch {args.global_batch})" if accum > 1 else ""),
                       "global_batch": args.batch * world * accum, "accumulate_grad_batches": accum,
                       "seq_len": args.seqlen, "parallelism": f"dp{world}",
                       "params": n_params, "final_loss": float(loss.detach())},
            "dist": dist_info, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
