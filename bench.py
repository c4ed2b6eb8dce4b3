#!/usr/bin/env python3
"""bench.py -- Msamples/s through demod_chain on MI355X (BASELINE.json metric), one process per GPU.

A step = one pass of the rank's demod_chains (modem -> slicer on the GPU, LFSR -> codec native on the host, packet
gather + de-dup) over one synthetic recording that is already resident in HBM.  Weak scaling: every rank runs
--chains-per-gpu chains, chains are independent (no data-path collective); the only exchange is the packet gather.
By default successive steps are pipelined the way a service decoding one recording after another would run them
(chain_execute.RecordingPipeline, --overlap 2): demod kernels, slicer streams and host stages of neighbouring steps overlap;
all K steps complete inside the timed region.  --overlap 0 times them strictly one after the other.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--samples S] [--chains-per-gpu C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 with `roofline` (dominant GPU kernel, HIP-event time measured inside the timed region
through pm_prof_*) and `cpu_baseline` (the oracle -- the CPU restatement of the reference -- on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

# Before anything loads the HIP runtime: eight hardware queues instead of the default four.  The executor keeps two demod and two slicer
# streams busy; with an RCCL communicator (its streams made first) and torch's own in the same process, four queues made those streams
# share queues and their kernels take turns -- the forced one-rank exchange ran at 1.02 ms per step against 0.77 without, and at
# 0.80 with eight queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the 8-tap FIR moves 6.3 TB/s, a plain copy kernel 4.7 (profiles/r01_ubench.txt)


# ---- workloads: BASELINE.json configs as lists of 'demod_chain' lines --------------------------------------------
def _afsk_line(name, space_gain, mark, space, span, codec="ax25"):
    opts = {"space_gain": str(space_gain), "mark_freq": str(mark), "space_freq": str(space)}
    if span is not None:
        opts["correlator_span"] = str(span)
    if codec == "ax25":
        stream, cdc = {"type": "lfsr", "options": {"poly": "0x3", "invert": "True"}}, {"type": "ax25"}
    else:
        stream = {"type": "lfsr", "options": {"poly": "0x1", "invert": "False"}}
        cdc = {"type": "il2p", "options": {"crc": "yes", "disable_rs": "no", "min_dist": "0", "sync_tol": "0"}}
    return {"object_name": name, "object_type": "demod_chain", "modem": {"type": "afsk", "config": "1200", "options": opts},
            "slicer": {"type": "binary", "config": "1200", "options": {"lock_rate": "0.77"}}, "stream": stream, "codec": cdc}


def wl_afsk_super_opt(j):
    """configs/afsk_1200_ax25_super_opt.json: chain 0 = 1600/1800 span 1.0; chains 1..7 = 1300/2100 span 1.5 with
    space_gain 1.25..2.75.  Beyond 8 chains (more GPUs) the space_gain sweep continues in finer steps."""
    k, rep = j % 8, j // 8
    if k == 0:
        return _afsk_line(f"AFSK 1200 AX.25 1600/1800 sg 1.0 #{rep}", 1.0 + 0.05 * rep, 1600.0, 1800.0, None)
    sg = 1.0 + 0.25 * k + 0.03 * rep
    return _afsk_line(f"AFSK 1200 AX.25 1300/2100 sg {sg:.2f}", sg, 1300.0, 2100.0, 1.5)


def wl_fsk_9600(j):
    """configs/fsk_9600.json: 3 chains sharing one 8-tap LPF (IL2P, IL2P inverted, G3RUH AX.25)."""
    k = j % 3
    il2p = {"type": "il2p", "options": {"crc": "yes", "disable_rs": "no", "min_dist": "0", "sync_tol": "2"}}
    stream = [{"poly": "0x1", "invert": "no"}, {"poly": "0x1", "invert": "yes"}, {"poly": "0x63003", "invert": "yes"}][k]
    return {"object_name": f"FSK 9600 #{j}", "object_type": "demod_chain", "modem": {"type": "fsk", "config": "9600", "options": {}},
            "slicer": {"type": "binary", "config": "9600", "options": {"lock_rate": "0.88"}},
            "stream": {"type": "lfsr", "options": stream}, "codec": il2p if k < 2 else {"type": "ax25"}}


def wl_bpsk_300(j):
    """configs/bpsk_300.json (single chain); replicas sweep the carrier."""
    return {"object_name": f"BPSK 300 IL2P+CRC {1500 + 5 * j}Hz", "object_type": "demod_chain",
            "modem": {"type": "bpsk", "config": "300", "options": {"carrier_freq": str(1500 + 5 * j)}},
            "slicer": {"type": "binary", "config": "300", "options": {"lock_rate": "0.90"}},
            "stream": {"type": "lfsr", "options": {"poly": "0x3", "invert": "True"}},
            "codec": {"type": "il2p", "options": {"crc": "yes", "disable_rs": "no", "min_dist": "0", "sync_tol": "2"}}}


def wl_qpsk_2400(j):
    """BASELINE configs[4]: replicated qpsk_2400 chains at swept tuning offsets (the bundled file sweeps 1475/1500/1525)."""
    step = (j + 1) // 2 * (1 if j % 2 else -1)      # 0, +1, -1, +2, -2, ...: every prefix of the 64-chain sweep is centred on the carrier
    f = 1500.0 + 3.125 * step
    return {"object_name": f"QPSK 2400 IL2P+CRC {f:g}", "object_type": "demod_chain",
            "modem": {"type": "mpsk", "config": "qpsk_2400", "options": {"carrier_freq": str(f)}},
            "slicer": {"type": "quadrature", "config": "qpsk_2400", "options": {"lock_rate": "0.98"}},
            "stream": {"type": "lfsr", "options": {"poly": "0x1", "invert": "False"}},
            "codec": {"type": "il2p", "options": {"crc": "yes", "disable_rs": "no", "min_dist": "0", "sync_tol": "2"}}}


WORKLOADS = {
    # name: (line factory, default chains/GPU, description)
    "afsk_1200_super_opt": (wl_afsk_super_opt, 8, "configs/afsk_1200_ax25_super_opt.json (BASELINE configs[3]): AFSK-1200 AX.25 chains, "
                            "BPF 148 + 4 correlators 40/60 + LPF 100 taps @48 kHz, binary slicer, NRZI, AX.25"),
    "fsk_9600": (wl_fsk_9600, 3, "configs/fsk_9600.json (BASELINE configs[2]): 8-tap LPF, binary slicer, IL2P / G3RUH AX.25"),
    "bpsk_300": (wl_bpsk_300, 1, "configs/bpsk_300.json (BASELINE configs[1]): BPF 240, AGC, Costas loop, RRC 961, binary slicer, IL2P"),
    "qpsk_2400": (wl_qpsk_2400, 8, "BASELINE configs[4]: qpsk_2400 MPSK chains at swept carriers (BPF 130, AGC, Hilbert 163, "
                  "carrier loop, 2x RRC 241, quadrature slicer, IL2P)"),
}

FP64_PEAK_TFLOPS = 78.6         # MI355X vector FP64 (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz); v_mfma_f64 has the same dense peak
FP64_SUSTAINED_TFLOPS = 64.3    # what a pure register loop of v_fma_f64 holds on every SIMD (tools/ubench/lat.hip, profiles/r01_ubench.txt)


SIGNAL_MODE = {"afsk_1200_super_opt": "afsk1200_ax25", "fsk_9600": "fsk9600_il2p", "bpsk_300": "bpsk300_il2p", "qpsk_2400": "qpsk2400_il2p"}
BUFFER_DESC = {"noise": "default_rng(1234).standard_normal(n)*8000 -> int16",
               "signal": "pymodem_amd.siggen packets of the workload's mode (seed 1234, amplitude 8000) + AWGN sigma 2000, one minute tiled to the recording length"}


def make_buffer(args, n=None):
    """The recording every chain of every rank processes (same seed on every rank)."""
    n = args.samples if n is None else n
    if args.buffer == "noise":
        return synth_buffer(n)
    from pymodem_amd import siggen
    minute = min(n, 60 * args.rate)
    mode = SIGNAL_MODE[args.workload]
    per_packet = {"afsk1200_ax25": 0.60, "fsk9600_il2p": 0.28, "bpsk300_il2p": 2.5, "qpsk2400_il2p": 0.45}[mode]   # lower bound, seconds incl. gap
    audio = np.zeros(0, np.int16)
    seed = 1234
    while len(audio) < minute:                    # noise everywhere, no digital silence: generate past the minute, then cut
        part, _ = siggen.recording(mode, args.rate, packets=max(1, int(minute / args.rate / per_packet)), seed=seed, noise_sigma=2000.0,
                                   payload_len=(20, 80))
        audio = np.concatenate([audio, part])
        seed += 1
    audio = audio[:minute]
    reps = -(-n // len(audio))
    return np.tile(audio, reps)[:n]


def synth_buffer(n, seed=1234):
    """BASELINE.md throughput buffer: default_rng(1234).standard_normal(n)*8000 -> int16."""
    x = np.random.default_rng(seed).standard_normal(n) * 8000.0
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="afsk_1200_super_opt", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=28_800_000, help="samples per recording (10 min @ 48 kHz)")
    ap.add_argument("--chains-per-gpu", type=int, default=0)
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", type=int, default=2,
                    help="how successive steps (recordings) overlap, as in a service decoding one recording after another. "
                         "2: three-stage pipeline (chain_execute.RecordingPipeline): demod of step k+1 on the default stream, slicer of "
                         "step k on a high-priority side stream, host half (LFSR, codec, packet gather, de-dup) of step k-1 in threads; "
                         "1: only the host half runs behind the next step's GPU half; 0: strictly one after the other")
    ap.add_argument("--executor", default="auto", choices=["auto", "native", "python"],
                    help="--overlap 2: native = the library's own pipelined executor (pm_pipe_*: one call per recording, slicer and host "
                         "stages on the library's threads); python = chain_execute.RecordingPipeline (the stages sequenced by Python "
                         "threads).  auto: native where it applies (AFSK gain-sweep configs), python elsewhere")
    ap.add_argument("--slice-workers", type=int, default=2, help="--overlap 2: recordings whose slicers may be in flight at once")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse ranks on one GPU)")
    ap.add_argument("--buffer", default="signal", choices=["signal", "noise"],
                    help="signal: seeded packet-bearing recording of the workload's mode + AWGN (pymodem_amd.siggen), tiled to --samples; "
                         "noise: BASELINE.md's default_rng(1234) noise buffer")
    ap.add_argument("--cpu-sample", type=int, default=0, help="samples for the CPU baseline leg (0 = auto)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --chains-per-gpu chains on EVERY rank (beyond the config's own chains the sweep continues in finer "
                         "steps, named in config.workload); strong: the config's own chains (8 for configs[3]) divided over the ranks")
    ap.add_argument("--loop-batch", type=int, default=0,
                    help="carrier-loop workloads (bpsk_300, qpsk_2400): recordings per engine run (pymodem_amd.loop_batch) -- the loops of all of "
                         "them x the rank's chains advance together, one lane each; a step is still one recording.  0 (default): as many as "
                         "give 16384 loops in flight (every lane of one stepping wave per CU), never more than --steps")
    ap.add_argument("--loop-chunk", type=int, default=0, help="carrier-loop workloads: final-filter outputs per time chunk (0: 65536 for runs of "
                    "more than 8192 loops, 131072 from 2048 recordings, else 262144 -- the work buffers are sized by it)")
    ap.add_argument("--also", type=int, default=1, help="1 (default, one GPU only): after the headline workload also measure fsk_9600, "
                    "bpsk_300 and qpsk_2400 (BASELINE configs[2], [1], [4]) briefly and attach them under 'also'")
    ap.add_argument("--with-exchange", type=int, default=1,
                    help="1 (default): a one-GPU line also carries value_with_exchange -- the same timed steps with the packet exchange "
                         "(one-rank RCCL gather + rank 0's de-dup over gathered rows) behind the executor, what every rank of an N > 1 run "
                         "does -- so that the first step of a 1 -> N curve compares like with like; an N > 1 weak-scaling line also carries "
                         "`strong`: the config's own chains divided over the ranks (one chain per GPU at N = 8)")
    args = ap.parse_args()

    if args.gpus > 1 and os.environ.get("RANK") is None:
        sys.exit(self_launch(args))

    # A dozen Python threads take turns here (the submitting thread, slicer workers, host, finish and post stages), each mostly inside
    # native calls that release the interpreter lock; with CPython's default 5 ms switch interval a thread coming back from a 20 us
    # native call can wait milliseconds for the lock.  Round 1: 0.5 ms (1.47 -> 1.33 ms per step).  With round 2's executor (fewer,
    # longer-running threads; the tiny calls no longer drop the lock at all) interleaved runs give, 20 steps / 400 steps, medians:
    # 0.5 ms 1.52 / 1.11, 1 ms 1.41 / 1.09, 2 ms 1.35 / 1.09, 5 ms 1.44 / 1.03 ms per step.
    sys.setswitchinterval(float(os.environ.get("BENCH_SWITCH_INTERVAL", "0.002")))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                             # the launcher's word: one process per GPU

    # CPU baseline first: it forks one process per chain, which must happen before this process initialises the GPU.  Rank 0 of every
    # run carries it (N > 1: the other ranks wait for rank 0 in the rendezvous meanwhile; a self-launched run has it from the parent).
    cpu_line, cpu_also = None, {}
    if rank == 0 and not args.no_cpu_baseline:
        handed = os.environ.get("BENCH_CPU_BASELINE_FILE")
        if handed and os.path.exists(handed):
            cpu_line = json.load(open(handed))
        else:
            factory, default_cpg, _ = WORKLOADS[args.workload]
            cpu_line = cpu_baseline(args, [factory(c) for c in range(args.chains_per_gpu or default_cpg)])
        if args.also and world == 1:                  # the same oracle leg, shorter, beside every `also` workload
            import copy
            for name in ALSO:
                if name != args.workload:
                    a = copy.copy(args)
                    a.workload, a.cpu_sample = name, ALSO_CPU_SAMPLE[name]
                    f2, cpg2, _ = WORKLOADS[name]
                    cpu_also[name] = cpu_baseline(a, [f2(c) for c in range(cpg2)], seconds=4.0, max_passes=4)

    # Libraries print banners on stdout (RCCL's version block when its first communicator comes up): everything but the JSON line
    # goes to stderr, the line itself to the real stdout.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL between processes needs on these hosts
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py needs a GPU")
    dev_index = local % ndev                      # more ranks than GPUs only happens in the gloo rehearsal
    torch.cuda.set_device(dev_index)
    use_dist = world > 1 or bool(os.environ.get("PYMODEM_AMD_FORCE_GATHER"))      # the latter: one rank, collectives still run

    def init_dist():
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            # the exchange is a tiny kernel on a GPU kept full by 14 000-workgroup FIR launches: let it jump the queue
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = os.environ.get("BENCH_NCCL_PRIO", "1") != "0"
            if os.environ.get("BENCH_NCCL_LAZY"):
                dist.init_process_group("nccl", pg_options=opts)
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), pg_options=opts)
        else:
            dist.init_process_group(args.backend)
    exchange_leg = world == 1 and not use_dist and bool(args.with_exchange) and args.overlap == 2
    comm_up = False
    if use_dist or (exchange_leg and os.environ.get("BENCH_EXCHANGE_COMM_FIRST", "1") != "0"):
        # (the one-GPU line's exchange leg: its communicator comes up HERE, before the first stream of the executor exists, as in every
        # rank of an N > 1 run -- made afterwards it leaves the exchange 30 % slower, profiles/r04_exchange_probe.txt; the headline steps
        # below do not use it)
        try:
            init_dist()
            comm_up = True
        except Exception:                                     # noqa: BLE001 -- the leg reports its own failure below
            if use_dist:
                raise
    coll_device = f"cuda:{dev_index}" if (use_dist and args.backend == "nccl") else None

    env = {"rank": rank, "world": world, "dev_index": dev_index, "use_dist": use_dist, "coll_device": coll_device}
    out = measure(args, env)
    if os.environ.get("BENCH_EX_SKIP") and out is not None:   # a diagnostic that drops part of the exchange: the line is not a measurement
        out["invalid"] = "BENCH_EX_SKIP=" + os.environ["BENCH_EX_SKIP"] + ": part of the packet exchange was skipped"
    import copy
    if exchange_leg:
        # The same steps once more with the one exchange step behind the executor, as every rank of an N > 1 run has it: rows packed
        # for the wire, ONE all_gather per 1-8 recordings (here among one rank, over RCCL), rank 0's de-dup over the gathered rows.
        try:
            if not comm_up:
                init_dist()
                comm_up = True
            a = copy.copy(args)
            a.no_cpu_baseline = True
            d2 = measure(a, dict(env, use_dist=True, coll_device=f"cuda:{dev_index}" if args.backend == "nccl" else None))
            out["value_with_exchange"] = d2["value"]
            out["with_exchange"] = {"value": d2["value"], "unit": d2["unit"], "ms_per_step": d2["ms_per_step"], "steps": d2["steps"],
                                    "steady_ms_per_step": (d2.get("steady_state") or {}).get("ms_per_step"), "packets": d2.get("packets"),
                                    "what": "the timed steps repeated with the packet exchange behind the executor (forced one-rank "
                                            f"{'RCCL' if args.backend == 'nccl' else args.backend} all_gather of the packed rows + rank 0's "
                                            "de-dup over them): the N = 1 point of a 1 -> N curve whose other points all exchange"}
        except Exception as e:                                # noqa: BLE001 -- never allowed to break the main line
            out["with_exchange"] = {"error": repr(e)[:300]}
    if world > 1 and args.scaling == "weak" and args.with_exchange:
        # north_star's own sharding -- the config's chains divided over the GPUs, one chain per GPU at N = 8 -- beside the weak-scaling
        # figure, same steps, same exchange (every rank runs it: the collectives inside must line up)
        a = copy.copy(args)
        a.scaling, a.no_cpu_baseline, a.chains_per_gpu = "strong", True, 0
        d3 = measure(a, env)
        if rank == 0:
            out["strong"] = {"value": d3["value"], "unit": d3["unit"], "ms_per_step": d3["ms_per_step"], "steps": d3["steps"],
                             "chains_total": d3["config"]["chains_total"], "parallelism": d3["config"]["parallelism"],
                             "packets": d3.get("packets"), "scaling": "strong"}
    if world > 1 and args.also:
        # BASELINE configs[4]'s own curve: the 64 chains of the qpsk_2400 sweep divided over the ranks (8 per GPU at N = 8, what
        # north_star names), every rank one engine run of as many recordings as give it 24 576 loops in flight, the packet exchange and
        # rank 0's de-dup over all 64 chains behind it.  Every rank runs it: the collectives inside must line up.  The N = 1 point of the
        # same curve is the one-GPU line's `also.qpsk_2400.all_64_chains_on_one_gpu`.
        a = copy.copy(args)
        a.workload, a.scaling, a.no_cpu_baseline, a.chains_per_gpu = "qpsk_2400", "strong", True, 64
        per_rank = -(-64 // world)
        rehearsal = {k: int(os.environ[k]) for k in ("BENCH_ALSO64_RECORDINGS", "BENCH_ALSO64_SAMPLES") if os.environ.get(k)}
        a.steps = rehearsal.get("BENCH_ALSO64_RECORDINGS") or max(1, LOOPS_IN_FLIGHT_TWO // per_rank)
        a.samples = rehearsal.get("BENCH_ALSO64_SAMPLES") or args.samples
        a.warmup, a.loop_batch, a.loop_chunk = 1, 0, 0
        d4 = measure(a, env)
        if rank == 0:
            out.setdefault("also", {})["qpsk_2400_64"] = {
                "value": d4["value"], "unit": d4["unit"], "ms_per_step": d4["ms_per_step"], "steps": d4["steps"], "n_gpus": world, "scaling": "strong",
                "chains_total": d4["config"]["chains_total"], "chains_per_gpu": d4["config"]["chains_per_gpu"],
                "samples_per_recording": a.samples, "parallelism": d4["config"]["parallelism"], "loop_batch": d4["config"].get("loop_batch"),
                "ms_per_step_by_rank": d4.get("ms_per_step_by_rank"), "packets": d4.get("packets"),
                "gpu_kernel_ms_per_step": d4.get("gpu_kernel_ms_per_step"),
                "what": "BASELINE configs[4]: 64 replicated qpsk_2400 chains at swept carriers over one synthetic recording per step, the chains "
                        "divided over the GPUs (contiguous blocks: one front end per GPU), one packet exchange + rank 0's de-dup over all 64",
                **({"rehearsal": rehearsal} if rehearsal else {})}
    if rank == 0:
        if cpu_line is not None:
            out["cpu_baseline"] = cpu_line
        if world == 1 and args.also:
            out["also"] = also_workloads(args, env, cpu_also)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())         # the ONE line of this run's stdout
    if comm_up:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


ALSO = ("fsk_9600", "bpsk_300", "qpsk_2400")
LOOP_WORKLOADS = ("bpsk_300", "qpsk_2400")
ALSO_CPU_SAMPLE = {"fsk_9600": 4_800_000, "bpsk_300": 1_440_000, "qpsk_2400": 1_440_000, "afsk_1200_super_opt": 4_800_000}


LOOP_DISTINCT_BUFFERS = 256    # carrier-loop workloads: distinct resident copies of the recording an engine run's recordings read (14.7 GB)
LOOPS_IN_FLIGHT = 16384        # carrier loops per engine run, one-output loops: 256 waves, eight on three of the loops' 96 CUs
LOOPS_IN_FLIGHT_TWO = 24576    # two-output loops (MPSK, QPSK): 384 waves, three on each of their 128 CUs -- what the sign bitmaps of whole
                               # recordings (118 GB at 16384 loops) used to forbid; measured 2048 / 2560 / 3072 / 3584 / 4096 recordings of 8
                               # chains per run: 5.10 / 4.40 / 4.43 / 4.47 / 4.59 ms per step (profiles/r04_loop_sweep.txt)


INT8_PEAK_TOPS = 5000.0         # MI355X dense int8 matrix peak: 2x the BF16 rate per clock (MI355X_MICROARCH.md, MFMA table), 2 ops per product


def matrix_roofline(native, args, modems, my, prof):
    """The native executor's certified sums as what they are since round 3: int8 products on the matrix pipe.  Algorithmic products per
    launch = taps x digit pairs x outputs, the digit pairs as the LIBRARY reports them from its kernels' own constants
    (pm_matrix_digit_pairs: 2 x 4 for the band-pass, 8 of 3 x 3 per low-pass stream since round 4) -- against the launch times of the
    same HIP-event profile (in the pipeline, beside the other stream's kernels and the slicers)."""
    try:
        if not native or args.workload != "afsk_1200_super_opt" or os.environ.get("PM_PIPE_LPF8") == "0":
            return None
        md = modems[my[0]]
        ml, mb, n = len(md.output_lpf), len(md.input_bpf), float(args.samples)
        sweeps = {}
        for c in my:
            key = modems[c].mark_key() if hasattr(modems[c], "mark_key") else c
            sweeps[key] = sweeps.get(key, 0) + 1
        streams = sum(2 if cnt > 1 else 1 for cnt in sweeps.values())            # low-pass streams per recording: two per sweep, one per lone chain
        from pymodem_amd import lib
        bp_pairs, lp_pairs = lib().pm_matrix_digit_pairs(0), lib().pm_matrix_digit_pairs(1)
        if bp_pairs <= 0 or lp_pairs <= 0:
            return {"error": "pm_matrix_digit_pairs"}
        lp_ops, bp_ops = 2.0 * lp_pairs * ml * n * streams, 2.0 * bp_pairs * mb * n
        out = {"bound": "mfma", "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "digit_pairs": {"band_pass": bp_pairs, "low_pass_per_stream": lp_pairs},
               "note": "int8 digit products of the certified sums (exact int32 accumulation, v_mfma_i32_16x16x64_i8), 2 ops per product, per "
                       "recording / the class's summed launch time per recording; a micro-benchmark of the instruction alone sustains 2 830 TOP/s "
                       "on this chip (tools/ubench/mfma_i8.hip); the kernels also run their sliding sums, digit split, recombination and "
                       "combine on the vector pipe"}
        fused = not prof["fir_i16"][1]                       # one launch per recording (afsk_fused8_kernel): band-pass and low-passes in the same class
        if fused:
            out["note"] += "; ONE fused launch per recording: the band-pass's and the low-passes' products are priced together against its time"
        for cls, ops in ((("fir_f64", lp_ops + bp_ops),) if fused else (("fir_f64", lp_ops), ("fir_i16", bp_ops))):
            ms, nl_ = prof[cls]
            if nl_ and ms > 0:
                per_rec_ms = ms / nl_ * (1 if fused else len(sweeps) if cls == "fir_f64" else 1)
                out[cls] = {"int8_ops_per_recording": round(ops), "class_ms_per_recording": round(per_rec_ms, 5),
                            "achieved": round(ops / (per_rec_ms * 1e-3) / 1e12, 1), "frac": round(ops / (per_rec_ms * 1e-3) / 1e12 / INT8_PEAK_TOPS, 5)}
        return out
    except Exception as e:                                    # noqa: BLE001  (an extra: never allowed to break the line)
        return {"error": repr(e)}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: this process -- which never touches the GPU -- runs the CPU
    baseline, then starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <the same arguments>` as a CHILD
    (never an exec: a process that holds the GPU must not be replaced, and this one may have forked already), relays rank 0's one
    JSON line and returns the child's exit code.  (pymodem.py:140-166: the reference's own one-process-per-chain launch.)"""
    import socket
    import subprocess
    import tempfile
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    tmp = None
    if not args.no_cpu_baseline:
        factory, default_cpg, _ = WORKLOADS[args.workload]
        cpu_line = cpu_baseline(args, [factory(c) for c in range(args.chains_per_gpu or default_cpg)])
        fd, tmp = tempfile.mkstemp(prefix="bench_cpu_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(cpu_line, f)
        env["BENCH_CPU_BASELINE_FILE"] = tmp
    with socket.socket() as sk:                               # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    try:
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        line = None
        for ln in child.stdout:                               # rank 0 prints ONE JSON line; anything else a library wrote goes to stderr
            t = ln.strip()
            if t.startswith("{") and '"metric"' in t:
                line = t
            elif t:
                print(t, file=sys.stderr)
        rc = child.wait()
    finally:
        if tmp and os.path.exists(tmp):
            os.unlink(tmp)
    if line is not None:
        print(line, flush=True)
    return rc if rc or line is not None else 1


def also_workloads(args, env, cpu_also=None):
    """BASELINE configs[1], [2] and [4] measured after the headline workload in the same process (one GPU only): the single-chain
    BPSK-300 Costas path and the 8-chain QPSK-2400 path are bound by their sequential carrier loops -- one step each
    at full size -- and fsk_9600 is the shortest-tap FIR path.  Never allowed to break the main line."""
    import copy
    out = {}
    # (the carrier-loop workloads: full engine runs -- 16384 recordings x 1 chain, 3072 x 8 chains; a run takes as long as
    # its recordings are, however many there are)
    # -- two runs each, one after the other: the host's share of the first lies beside the GPU's run of the second (loop_steps)
    for name, steps, warm in (("fsk_9600", 300, 10), ("bpsk_300", 32768, 1), ("qpsk_2400", 6144, 1)):
        if name == args.workload or (os.environ.get("BENCH_ALSO_ONLY") and name not in os.environ["BENCH_ALSO_ONLY"].split(",")):
            continue
        a = copy.copy(args)
        a.workload, a.steps, a.warmup, a.no_cpu_baseline, a.chains_per_gpu = name, steps, warm, True, 0
        a.loop_batch, a.loop_chunk = 0, 0
        try:
            d = measure(a, env)
            out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": steps, "config": d["config"]["workload"],
                         "chains_per_gpu": d["config"]["chains_per_gpu"], "samples_per_recording": a.samples,
                         "gpu_kernel_ms_per_step": d["gpu_kernel_ms_per_step"], "packets": d["packets"],
                         "roofline_kernel": d["roofline"]["kernel"], "roofline_frac_alone": (d["roofline"]["alone"] or {}).get("frac"),
                         "roofline_fp64_alone_frac": d["roofline_fp64"]["alone_frac"], "dominant_by_time": d["roofline"]["dominant_by_time"]}
            if cpu_also and name in cpu_also:
                out[name]["cpu_baseline"] = cpu_also[name]
            if d["config"].get("loop_batch"):
                out[name]["loop_batch"] = d["config"]["loop_batch"]
            if name == "qpsk_2400":
                # configs[4] has 64 chains: the whole config on ONE GPU, 384 recordings in flight (the same 24576 loops per launch as 3072
                # recordings x 8 chains), for comparison with the 8-chains-per-GPU sharding
                a64 = copy.copy(a)
                a64.chains_per_gpu, a64.steps, a64.warmup, a64.loop_batch, a64.loop_chunk = 64, 384, 1, 0, 0
                d64 = measure(a64, env)
                out[name]["all_64_chains_on_one_gpu"] = {"value": d64["value"], "unit": d64["unit"], "ms_per_step": d64["ms_per_step"], "chains_per_gpu": 64,
                                                         "steps": 384, "loop_batch": d64["config"]["loop_batch"],
                                                         "gpu_kernel_ms_per_step": d64["gpu_kernel_ms_per_step"], "packets": d64["packets"]}
        except Exception as e:                                   # noqa: BLE001
            out.setdefault(name, {})["error"] = repr(e)[:300]
    return out


def measure(args, env):
    """One workload: build the chain group, W warm-up steps, K timed steps -> the dict of the JSON line (rank 0) or None."""
    import torch
    rank, world, dev_index, use_dist, coll_device = env["rank"], env["world"], env["dev_index"], env["use_dist"], env["coll_device"]
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce, dist as pdist
    ctx = pymodem_amd.Context.default(dev_index)  # the stage objects use the same per-process default context

    factory, default_cpg, desc = WORKLOADS[args.workload]
    cpg = args.chains_per_gpu or default_cpg
    if args.scaling == "strong":                                  # the config's own chains divided over the ranks (contiguous blocks)
        nchains = cpg
        my = [c for c in range(nchains) if c * world // nchains == rank] if world <= nchains else ([rank] if rank < nchains else [])
    else:
        nchains = cpg * world
        my = [c for c in range(nchains) if c // cpg == rank]      # contiguous blocks: chains sharing a front end stay together
    lines = {c: factory(c) for c in range(nchains)}
    names = [lines[c]["object_name"] for c in range(nchains)]

    audio = make_buffer(args)
    d_audio = ctx.upload(audio)                                   # resident in HBM before the timed region
    ctx.sync()
    ring = [d_audio]

    def audio_ring(count):
        """`count` DISTINCT resident copies of the recording (made outside the timed region): successive recordings read different
        memory, as a service decoding different recordings does -- one buffer submitted over and over stays in L2 / Infinity Cache
        for whoever comes next.  More than the executor has recordings in flight, so none is still cached when its turn
        comes again."""
        from pymodem_amd._native import check, lib
        while len(ring) < count:
            b = ctx.empty(len(audio), np.int16)
            check(lib().pm_d2d(ctx.handle, b.ptr, d_audio.ptr, audio.nbytes))
            ring.append(b)
        ctx.sync()
        return ring[:count]
    modems = {c: cb.ModemConfigurator(args.rate, lines[c]["modem"]) for c in my}     # tap design once (host)

    chains_ref = []

    def build_chains(reset=True):
        chains = []
        for c in my:
            line = lines[c]
            modem = modems[c]
            if reset:
                modem.reset()
            srate = getattr(modem, "output_sample_rate", args.rate)
            chains.append([line["object_name"], modem, cb.SlicerConfigurator(srate, line["slicer"]),
                           cb.StreamConfigurator(line["stream"]), cb.CodecConfigurator(line["codec"], line["object_name"])])
        chains_ref[:] = chains
        return chains

    def exchange(rows_list):
        return pdist.exchange_rows(dict(zip(my, rows_list)), nchains, device=coll_device)         # the one exchange step (collective)

    def dedupe(x):
        table = pdist.table_from_exchange(x, names)
        return table.correlate(args.rate / 40) if table is not None else None                      # rank 0: cross-chain de-dup

    def finish(rows_list):
        return dedupe(exchange(rows_list))

    def step():
        return finish(ce.process_chains_split(build_chains(), d_audio)())

    stage_ms = {}
    steady = {}
    pipes = {}
    # The carrier-loop workloads (BASELINE configs[1] and [4]): a step is still one recording, but up to --loop-batch recordings go
    # through ONE engine run -- the sequential carrier loop of every recording x chain on a lane of its own, all of them advancing
    # together in time chunks (pymodem_amd.loop_batch) -- then all slicers in batches and LFSR + codec per recording.
    loop_wl = args.workload in LOOP_WORKLOADS
    loop_info = None
    if loop_wl:
        from pymodem_amd import loop_batch as lb
        in_flight = LOOPS_IN_FLIGHT_TWO if args.workload in ("qpsk_2400",) and os.environ.get("PYMODEM_AMD_LOOP_FUSED_SLICERS", "1") != "0" else LOOPS_IN_FLIGHT
        batch = max(1, min(args.loop_batch or min(16384, in_flight // max(len(my), 1)), max(args.steps, 1)))
        if not args.loop_chunk:
            args.loop_chunk = 65536 if batch * len(my) > 8192 else 131072 if batch >= 2048 else 262144
        engine = lb.engine_for([modems[c] for c in my], batch, ctx, args.loop_chunk)
        if os.environ.get("PYMODEM_AMD_LOOP_FUSED_SLICERS", "1") != "0":      # the slicers inside the engine: rows of bytes, no bitmaps
            for sl_ in range(2 if args.steps > batch else 1):     # (two sets of output rows when batches follow each other: see loop_steps)
                engine.reserve_sliced(batch, args.samples, [ch[2] for ch in build_chains(reset=False)], slot=(sl_, 0))
        else:
            engine.reserve(batch, args.samples, slot=(0, 0))
        nout_, chunk_, chunks_ = engine.geometry(args.samples)
        loop_info = {"recordings_per_run": batch, "runs": -(-max(args.steps, 1) // batch), "loops_in_flight": batch * len(my), "chunk_outputs": chunk_,
                     "chunks_per_recording": chunks_,
                     "slicers": "inside the engine, one lane per stream (pm_lbatch_run_sliced)" if os.environ.get("PYMODEM_AMD_LOOP_FUSED_SLICERS", "1") != "0"
                     else "pm_slice_batch over the run's sign bitmaps",
                     "between_runs": "the host's share of a run (rows to the host, LFSR + codec, de-dup) beside the GPU's next run; the phase times "
                                     "below then overlap and add up to more than the step",
                     "note": "all carrier loops of the run's recordings x this rank's chains advance together, one lane each, state carried "
                             "in device memory from chunk to chunk; band-pass/AGC/Hilbert of chunk t+1 on a second stream beside the loops of chunk t"}

    loop_phase_s = {}

    def loop_steps(k, audio_dev):
        import threading
        res = [None]
        loop_phase_s.clear()

        failed = []

        def after(rest, st):
            """what follows a batch's engine run: its rows to the host, LFSR + codec, de-dup"""
            try:
                after_(rest, st)
            except BaseException as e:                            # noqa: BLE001  (raised again on the submitting thread)
                failed.append(e)

        def after_(rest, st):
            rows = rest()
            t_f = time.perf_counter()
            if use_dist and len(rows) > 8:
                # A run's recordings through the exchange as the pipelined executor's go: packed rows handed to dist.Exchanger in order on
                # THIS thread (the collectives must come in the same order on every rank), eight recordings per all_gather, two
                # collectives outstanding, and rank 0's indexing + de-dup of the gathered streams on four threads beside them
                # (one collective and one de-dup per recording, one after the other: 3072 x ~3 ms behind a 14-second engine run)
                from concurrent.futures import ThreadPoolExecutor
                ex = pdist.Exchanger(nchains, coll_device, batch=int(os.environ.get("PYMODEM_AMD_EXCHANGE_BATCH", "8")))
                ex.depth = 2
                with ThreadPoolExecutor(max_workers=4) as fin:
                    pending = []
                    for rr in rows:
                        pending.append(fin.submit(lambda f: dedupe(f.result()), ex.step(dict(zip(my, rr)))))
                        while len(pending) > 24 and pending[0].done():
                            res[0] = pending.pop(0).result()
                    ex.flush()
                    for p_ in pending:
                        res[0] = p_.result()
            elif use_dist or len(my) < 4:
                for rr in rows:                                   # (collectives: in order, on this thread; one chain: nothing to share out)
                    res[0] = finish(rr)
            else:
                # the recordings' de-dups do not depend on each other (PacketTable + pm_correlate, mostly native): four at a time
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=4) as fin:
                    for got in fin.map(finish, rows):
                        res[0] = got
            for k2, v in dict(st.get("seconds", {}), finish=time.perf_counter() - t_f).items():
                loop_phase_s[k2] = loop_phase_s.get(k2, 0.0) + v

        # Successive batches as a service would run them: the host's share of batch i (rows to the host, LFSR + codec, de-dup) on a
        # thread of its own beside the GPU's run of batch i + 1 (two sets of output rows in rotation).  One batch: nothing to overlap.
        # With an exchange the finish is a collective: in order, on this thread.
        prev = None
        starts = list(range(0, k, batch))
        built = {}

        def build(bi):                                            # a batch's stage objects, made while the batch before it runs
            built[bi] = [build_chains(reset=False) for _ in range(min(batch, k - starts[bi]))]
        build(0)
        for bi, b0 in enumerate(starts):
            r = min(batch, k - b0)
            sets = built.pop(bi)
            ahead = None
            if bi + 1 < len(starts):
                ahead = threading.Thread(target=build, args=(bi + 1,), name="bench-build-chains")
                ahead.start()
            st = {}
            bufs = audio_ring(min(r, LOOP_DISTINCT_BUFFERS))      # (16384 distinct ten-minute recordings would be 944 GB: each copy serves r / 256 of the run's recordings)
            rest = lb.process_recordings_device(sets, [bufs[i % len(bufs)] for i in range(r)], ctx, chunk=args.loop_chunk, rows=True, chain_ids=my, stages=st,
                                                slot=bi & 1, defer=True)
            if prev is not None:
                prev.join()
            if ahead is not None:
                ahead.join()
            if failed:
                raise failed[0]
            if use_dist:
                after(rest, st)
            else:
                prev = threading.Thread(target=after, args=(rest, st), name="bench-after-run")
                prev.start()
        if prev is not None:
            prev.join()
        if failed:
            raise failed[0]
        return res[0]

    native_sides = []
    exchanging = use_dist or bool(os.environ.get("PYMODEM_AMD_FORCE_GATHER"))

    def native_pipe(key):
        """The library's own executor for this rank's chains (made once, kept across the warm-up and the timed call)."""
        npipe = pipes.get(key)
        if npipe is None:
            # (with an exchange behind it the de-dup is rank 0's, over every rank's rows: none inside the executor)
            npipe = pipes[key] = ce.NativePipeline(build_chains(), args.samples, -1.0 if exchanging else args.rate / 40, ctx=ctx,
                                                   names=[names[c] for c in my], chain_ids=my, slice_workers=args.slice_workers)
            native_sides[:] = npipe.side_contexts()
            sides.extend(native_sides)
        return npipe

    def native_steps(npipe, k, source, pace=True):
        """k recordings through the native executor; every recording's result is taken (its de-dup count read, its rows given back),
        the last one's comes back as the PacketTable the Python executor's `dedupe` returns."""
        if exchanging:
            return native_steps_exchange(npipe, k, source)
        tickets, taken = [], 0
        nxt = npipe.prefetch(source) if (k and not hasattr(source, "ptr")) else None
        bufs = audio_ring(npipe.slots + 2) if hasattr(source, "ptr") else None
        for i in range(k):
            if nxt is not None:
                cur, nxt = nxt, (npipe.prefetch(source) if i + 1 < k else None)
            else:
                cur = bufs[i % len(bufs)]
            tickets.append(npipe.submit(cur))
            while taken < len(tickets) - 48:                  # results do not pile up: 7 MB of packet rows each
                npipe.unique(tickets[taken])
                taken += 1
        _tt = [time.perf_counter()]
        for t in tickets[taken:-1]:
            npipe.unique(t)
            _tt.append(time.perf_counter())
        res = npipe.table(tickets[-1]) if tickets else None
        _tt.append(time.perf_counter())
        npipe.drain()
        _tt.append(time.perf_counter())
        if os.environ.get("BENCH_TAIL_TRACE"):
            print("[tail] submits done -> each unique -> table -> drain (ms):", [round((x - _tt[0]) * 1e3, 3) for x in _tt], file=sys.stderr)
        # the pace between the 6th recording done and the 5th-from-last: the steady state, without fill and drain (as for the Python executor)
        done = sorted(npipe.done_at_ms.pop(t) for t in tickets if t in npipe.done_at_ms)
        if pace:
            steady.clear()
            if len(done) >= 16:
                steady["ms_per_step"] = round((done[-6] - done[5]) / (len(done) - 11), 4)
                steady["steps"] = len(done) - 11
        return res

    def native_steps_exchange(npipe, k, source):
        """The same with the one exchange step behind the executor (N > 1, or the forced one-rank exchange): every recording's rows are
        packed for the wire (two threads), handed to dist.Exchanger.step by ONE thread in submission order -- the collectives must come in
        the same order on every rank -- and de-duplicated on rank 0 by a third."""
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        # eight recordings per collective: the ordered thread's half-dozen torch calls and rank 0's indexing + de-dup cost about the same per
        # CALL whatever they carry (forced one-rank RCCL exchange, 400 steps, round 3: 1.31 ms per step with one recording per collective,
        # 1.06-1.08 with four or eight, 0.93 without the exchange; round 4, with the executor at 0.65-0.66: 0.80 with four, 0.69 with eight --
        # and 0.66 with rank 0's de-dup skipped, 0.65 with the collective skipped: profiles/r04_exchange_probe.txt)
        # (short runs -- the driver's 20 steps -- one per collective and one outstanding: what is gained per call there is lost in the
        # drain, 1.48-1.50 against 1.50-1.62 ms per step)
        ex = pdist.Exchanger(nchains, coll_device, batch=int(os.environ.get("PYMODEM_AMD_EXCHANGE_BATCH", "8" if k >= 64 else "1")))
        if "PYMODEM_AMD_EXCHANGE_DEPTH" not in os.environ:
            ex.depth = 2 if k >= 64 else 1
        packed, gathered, out, errors = queue.Queue(), queue.Queue(), {}, []

        acc = {"pack": 0.0, "step": 0.0, "dedupe": 0.0, "wait_rows": 0.0}

        def pack(t):
            t0_ = time.perf_counter()
            rows_ = npipe.rows(t)
            t1_ = time.perf_counter()
            if os.environ.get("BENCH_EX_SKIP") == "all":
                return None
            out_ = ex.prepare(dict(zip(my, rows_)))
            acc["wait_rows"] += t1_ - t0_
            acc["pack"] += time.perf_counter() - t1_
            return out_

        def timed_dedupe(g):
            x_ = g.result()
            t0_ = time.perf_counter()
            if os.environ.get("BENCH_EX_SKIP") == "dedupe":
                return None
            r_ = dedupe(x_)
            acc["dedupe"] += time.perf_counter() - t0_
            return r_

        def ordered():
            try:
                while True:
                    f = packed.get()
                    if f is None:
                        break
                    x_ = f.result()
                    if x_ is None or os.environ.get("BENCH_EX_SKIP") == "step":
                        continue
                    t0_ = time.perf_counter()
                    gathered.put(ex.step(x_))
                    acc["step"] += time.perf_counter() - t0_
                ex.flush()
            except BaseException as e:                        # noqa: BLE001
                errors.append(e)
            gathered.put(None)

        def post():
            # rank 0 indexes and de-duplicates every rank's rows: 2 ms per recording at 64 chains, more than a step -- several at a time
            # (the recordings' results do not depend on each other; the last one's is what the caller gets)
            try:
                with ThreadPoolExecutor(max_workers=4) as dedupers:
                    pending = []
                    while True:
                        f = gathered.get()
                        if f is None:
                            break
                        pending.append(dedupers.submit(timed_dedupe, f))
                        while len(pending) > 8:
                            pending.pop(0).result()
                    for p_ in pending[:-1]:
                        p_.result()
                    if pending:
                        out["last"] = pending[-1].result()
            except BaseException as e:                        # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=ordered), threading.Thread(target=post)]
        for th in threads:
            th.start()
        with ThreadPoolExecutor(max_workers=int(os.environ.get("BENCH_PACKERS", "2"))) as packers:
            nxt = npipe.prefetch(source) if (k and not hasattr(source, "ptr")) else None
            if nxt is None and k and os.environ.get("BENCH_SUBMIT_MANY", "1") != "0":
                # resident recordings: all of them from one library call on its own thread -- the submitting thread no longer queues
                # for the interpreter lock behind the packers, the collectives and rank 0's de-dup after every recording
                bufs = audio_ring(npipe.slots + 2)
                first, join = npipe.submit_many([bufs[i % len(bufs)] for i in range(k)])
                for i in range(k):
                    packed.put(packers.submit(pack, first + i))
                join()
                k_loop = 0
            else:
                k_loop = k
            for i in range(k_loop):
                if nxt is not None:
                    cur, nxt = nxt, (npipe.prefetch(source) if i + 1 < k else None)
                else:
                    bufs = audio_ring(npipe.slots + 2)
                    cur = bufs[i % len(bufs)]
                t0_ = time.perf_counter()
                tk_ = npipe.submit(cur)
                t1_ = time.perf_counter()
                packed.put(packers.submit(pack, tk_))
                acc["submit_call"] = acc.get("submit_call", 0.0) + t1_ - t0_
                acc["submit_rest"] = acc.get("submit_rest", 0.0) + time.perf_counter() - t1_
            packed.put(None)
            t_sub = time.perf_counter()
            if os.environ.get("BENCH_EXCHANGE_TAIL") and k and "submit_call" in acc:
                print(f"[exchange] k={k}: per step the submitting thread spent {1e3 * acc['submit_call'] / k:.3f} ms in pm_pipe_submit (slots) and "
                      f"{1e3 * acc['submit_rest'] / k:.3f} ms handing the ticket to a packer", file=sys.stderr)
            npipe.drain()
            t_pipe = time.perf_counter()
            for th in threads:
                th.join()
            if os.environ.get("BENCH_EXCHANGE_TAIL"):
                print(f"[exchange] k={k}: last submit to pipeline drained {1e3 * (t_pipe - t_sub):.2f} ms, then exchange + de-dup tail "
                      f"{1e3 * (time.perf_counter() - t_pipe):.2f} ms", file=sys.stderr)
        npipe.drain()
        if errors:
            raise errors[0]
        pdist.prewarm_wire_blocks(nchains, 24, 4)             # (after the warm-up call: the timed one finds its blocks made)
        if k:
            stage_ms.update({"exchange_pack": round(acc["pack"] / k * 1e3, 3), "exchange_ordered_step": round(acc["step"] / k * 1e3, 3),
                             "exchange_dedupe_rank0": round(acc["dedupe"] / k * 1e3, 3)})
        return out.get("last")

    def run_steps(k):
        """k steps; with --overlap the host half of each step runs behind the GPU half of the next one."""
        if loop_wl:
            return loop_steps(k, d_audio)
        if native_exec[0] and args.overlap >= 2:
            npipe = native_pipe("native")
            before = npipe.stats()
            stage_ms.clear()
            res = native_steps(npipe, k, d_audio)
            after = npipe.stats()
            if k:
                stage_ms.update({"executor": "native (pm_pipe_*)", "slice_busy": round((after["slice_busy_ms"] - before["slice_busy_ms"]) / k, 3),
                                 "host_busy": round((after["host_busy_ms"] - before["host_busy_ms"]) / k, 3),
                                 "recordings_per_slice_batch": round((after["recordings"] - before["recordings"]) / max(after["slice_batches"] - before["slice_batches"], 1), 2)})
            return res
        if not args.overlap:
            res = None
            for _ in range(k):
                res = step()
            return res
        if args.overlap >= 2:
            # the executor (its threads, streams and work blocks) lives across calls, like the process that owns it would keep it;
            # the timed region ends when every one of its k recordings has left the last stage (drain)
            pipe = pipes.get("main")
            if pipe is None:
                pipe = pipes["main"] = ce.RecordingPipeline(slice_workers=args.slice_workers)
            pipe.reset_stats()
            # the exchange runs one recording behind (dist.Exchanger): the copy back of a gather never waits for the collective
            ex = pdist.Exchanger(nchains, coll_device)
            # one rank and no forced exchange: nothing behind the host stage has to keep the recordings' order
            local_only = not use_dist and not os.environ.get("PYMODEM_AMD_FORCE_GATHER") and not os.environ.get("BENCH_ORDERED_TAIL")
            last = None
            for _ in range(k):
                last = pipe.submit(build_chains(), d_audio, ex.step, (None if os.environ.get("BENCH_NO_POST") else dedupe), prepare=lambda rows_list: ex.prepare(dict(zip(my, rows_list))), chain_ids=my, unordered=local_only)
            pipe.flush_finish(ex.flush)
            res = last.result() if last is not None else None
            pipe.drain()                                      # every step's de-dup is done, not only the last one's
            if os.environ.get("BENCH_FRESH_PIPE"):            # diagnostic: a new executor per call, torn down inside the timed region
                pipes.pop("main").close()
            stage_ms.clear()
            stage_ms.update({s: round(v / max(k, 1) * 1e3, 3) for s, v in pipe.stage_seconds.items()})
            # when each recording left its last stage: the pace between the 6th and the 5th-from-last is the pipeline's steady state,
            # without the fill (first demods alone on the GPU) and the drain (the last batch, its host stage, de-dup)
            done = sorted(max(v for kk, v in rec.items() if kk in ("post1", "finish1", "host1")) for rec in pipe.timeline if "host1" in rec)
            steady.clear()
            if len(done) >= 16:
                steady["ms_per_step"] = round((done[-6] - done[5]) / (len(done) - 11) * 1e3, 4)
                steady["steps"] = len(done) - 11
            if os.environ.get("BENCH_TIMELINE"):              # diagnostic: host clock at every stage boundary of every recording
                t00 = pipe.timeline[0]["submit0"] if pipe.timeline else 0
                for i, rec in enumerate(pipe.timeline):
                    print("[timeline]", i, {k2: round((v - t00) * 1e3, 2) for k2, v in rec.items()}, file=sys.stderr)
                print("[timeline] end of run_steps", round((time.perf_counter() - t00) * 1e3, 2), file=sys.stderr)
            if os.environ.get("BENCH_SLICE_LOG"):
                t00 = pipe.slice_log[0][1] if pipe.slice_log else 0
                print("[slice batches] (recordings, start ms, duration ms):", [(n, round((t - t00) * 1e3, 1), round(d * 1e3, 1)) for n, t, d in pipe.slice_log][:60], file=sys.stderr)
            return res
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as worker:       # one worker: host halves (and their collectives) stay in step order
            pending = None
            for _ in range(k):
                host_half = ce.process_chains_split(build_chains(), d_audio)
                if pending is not None:
                    pending.result()
                pending = worker.submit(lambda h=host_half: finish(h()))
            return pending.result() if pending is not None else None

    sides = [pymodem_amd.Context.side(dev_index, i) for i in range(args.slice_workers)] if args.overlap >= 2 else []
    if args.overlap >= 2 and int(os.environ.get("PYMODEM_AMD_CU_SPLIT", "0")) > 0:
        sides.append(pymodem_amd.Context.side(dev_index, 101, high_priority=False))       # the demod stream on the CUs the slicers leave

    def run_steps_uploading(k):
        """The same pipeline, but every step's recording starts in host memory: its copy to HBM runs one step ahead on a copy stream.
        The source is page-locked (registered once, below): out of pageable memory the runtime stages the copy in pieces at ~45 GB/s,
        which bounded this figure in round 2."""
        if native_exec[0]:
            return native_steps(native_pipe("native-upload"), k, audio, pace=False)
        pipe = pipes.get("upload")                            # kept across the warm-up and the timed call, like the main one
        if pipe is None:
            pipe = pipes["upload"] = ce.RecordingPipeline(slice_workers=args.slice_workers)
        ex = pdist.Exchanger(nchains, coll_device)
        local_only = not use_dist and not os.environ.get("PYMODEM_AMD_FORCE_GATHER") and not os.environ.get("BENCH_ORDERED_TAIL")
        last, nxt = None, (pipe.prefetch(audio) if k else None)
        for i in range(k):
            cur, nxt = nxt, (pipe.prefetch(audio) if i + 1 < k else None)
            last = pipe.submit(build_chains(), cur, ex.step, dedupe, prepare=lambda rows_list: ex.prepare(dict(zip(my, rows_list))), chain_ids=my,
                               unordered=local_only)
        pipe.flush_finish(ex.flush)
        res = last.result() if last is not None else None
        pipe.drain()
        return res

    native_exec = [False]
    if args.executor != "python" and args.overlap >= 2 and not loop_wl:
        try:
            native_pipe("native")
            native_exec[0] = True
        except ValueError:                                    # not a gain-sweep config: the Python-sequenced executor
            if args.executor == "native":
                raise
    elif args.executor == "native":
        raise SystemExit("--executor native: --overlap 2 and an AFSK gain-sweep workload")

    def close_pipes():
        for sc in native_sides:                               # the library's own contexts go with their pipeline
            if sc in sides:
                sides.remove(sc)
        del native_sides[:]
        for p_ in pipes.values():
            p_.close()
        pipes.clear()

    def fence():
        ctx.sync()
        for sc in sides:
            sc.sync()
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()

    if loop_wl:
        # any run of the engine takes as long as its recordings are (the loops are sequential in time): warm up on the first seconds
        if args.warmup:
            loop_steps(min(args.warmup, batch), d_audio.view(0, min(args.samples, 1_500_000)))
        sides.extend([engine.front, engine.tail, engine.slicing] + ([engine.loop] if engine.loop is not None else []))
    else:
        run_steps(args.warmup)
    fence()
    # The cyclic collector stops every thread (it runs under the interpreter lock): a full collection over the millions of objects
    # torch and numpy bring along took 7-15 ms and landed inside about one 20-step run in three (two slicer batches and the submitting
    # thread stalled together in the per-recording timeline).  Everything alive after the warm-up goes to the permanent generation;
    # what the steps allocate is still collected, in collections that only have those objects to look at.
    import gc
    gc.collect()
    gc.freeze()
    if not os.environ.get("BENCH_NO_PROF"):
        ctx.profile(True)
        for sc in sides:
            sc.profile(True)
    if os.environ.get("BENCH_TIMELINE"):
        print("[timeline] timed region begins", file=sys.stderr, flush=True)
    def thread_cpu():
        """{tid: (name, cpu seconds)} of every thread of this process (diagnostic: BENCH_THREAD_CPU=1)"""
        out = {}
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                st = open(f"/proc/self/task/{tid}/stat").read()
                name = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tick)
            except OSError:
                pass
        return out
    tcpu0 = thread_cpu() if os.environ.get("BENCH_THREAD_CPU") else None
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    if os.environ.get("BENCH_PYPROFILE") == "all":            # every thread's interpreter time (diagnostic, stderr): one profiler per thread
        import cProfile, pstats, threading
        profs = []

        def boot(*_a):
            pr = cProfile.Profile()
            profs.append(pr)
            sys.setprofile(None)
            pr.enable()
        close_pipes()                                         # threads are made when the executor is: start a fresh one under the hook
        threading.setprofile(boot)
        pr0 = cProfile.Profile()
        result = pr0.runcall(run_steps, args.steps)
        threading.setprofile(None)
        st = pstats.Stats(pr0, stream=sys.stderr)
        for pr in profs:
            try:
                pr.create_stats()
                st.add(pr)
            except Exception:                                 # noqa: BLE001
                pass
        st.sort_stats("tottime").print_stats(int(os.environ.get("BENCH_PYPROFILE_LINES", 45)))
    elif os.environ.get("BENCH_PYPROFILE"):                   # where the submitting thread's time goes (diagnostic, stderr)
        import cProfile, pstats
        pr = cProfile.Profile()
        result = pr.runcall(run_steps, args.steps)
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(28)
    else:
        result = run_steps(args.steps)
    _t1 = time.perf_counter()
    fence()
    elapsed = time.perf_counter() - t0
    if os.environ.get("BENCH_TAIL_TRACE"):
        print(f"[tail] run_steps {(_t1 - t0) * 1e3:.3f} ms, fence {(time.perf_counter() - _t1) * 1e3:.3f} ms", file=sys.stderr)
    if os.environ.get("BENCH_TIMELINE"):
        print("[timeline] timed region ends", file=sys.stderr, flush=True)
    host_cpu_ms = (time.process_time() - cpu0) / max(args.steps, 1) * 1e3     # all threads of this process, native ones included
    if tcpu0 is not None:                                     # which threads the host CPU of a step goes to, by thread name
        now, by = thread_cpu(), {}
        for tid, (name, cpu) in now.items():
            d = cpu - tcpu0.get(tid, (name, 0.0))[1]
            if d > 0:
                by.setdefault(name, [0, 0.0])
                by[name][0] += 1
                by[name][1] += d
        print("[thread cpu] ms per step by thread name (threads): " +
              ", ".join(f"{k} {v[1] / max(args.steps, 1) * 1e3:.2f} ({v[0]})" for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])), file=sys.stderr)
    if os.environ.get("BENCH_NO_PROF"):                   # diagnostic: the step time without the HIP-event bracketing the roofline needs
        print(json.dumps({"ms_per_step_without_profiling": round(elapsed / args.steps * 1e3, 3), "stages": stage_ms}), file=sys.stderr)
        sys.exit(0)
    prof = ctx.profile_read()
    work = ctx.profile_work()
    # when every launch of the FIR/correlator classes began and ended, over all streams (the native executor demodulates on three):
    # how many launches of a class were in flight together, and for how long the class kept the GPU busy at all
    spans = {}
    for name in ("fir_i16", "fir_f64", "afsk_correlate"):
        iv = [c.profile_intervals(name) for c in [ctx] + sides]
        iv = np.concatenate([v for v in iv if len(v)]) if any(len(v) for v in iv) else np.zeros((0, 2))
        if len(iv):
            iv = iv[np.argsort(iv[:, 0])]
            ends = np.maximum.accumulate(iv[:, 1])
            busy = float(np.sum(ends - np.maximum(iv[:, 0], np.concatenate(([iv[0, 0]], ends[:-1])))))      # length of the union
            spans[name] = {"busy_ms": busy, "sum_ms": float(np.sum(iv[:, 1] - iv[:, 0])), "launches": int(len(iv))}
    ctx.profile(False)
    for sc in sides:
        for name, (ms, cnt) in sc.profile_read().items():
            prof[name] = (prof[name][0] + ms, prof[name][1] + cnt)
        for name, (by, fl) in sc.profile_work().items():
            work[name] = (work[name][0] + by, work[name][1] + fl)
        sc.profile(False)
    # the same kernels with the GPU to themselves (one more step, strictly sequential, outside the timed region): in the pipelined
    # run two or more streams share the CUs, which stretches every kernel's duration
    alone = {}
    if native_exec[0] and args.overlap >= 2 and not loop_wl and not exchanging and args.steps > 1:
        # The native executor's own kernels with the GPU to themselves: the same pipeline, ONE recording in flight -- submitted, and its
        # result taken (demod, slicers and host stage drained) before the next is submitted.  Its demod stage then runs on one stream with
        # nothing beside it: the matrix-pipe kernels the timed region ran (round 4 timed the Python-sequenced path's binary64 kernels here).
        npipe = native_pipe("native")
        bufs = audio_ring(npipe.slots + 2)

        def one_alone(i):
            npipe.unique(npipe.submit(bufs[i % len(bufs)]))
        for i in range(2):                                    # both demod streams once, unprofiled
            one_alone(i)
        fence()
        watched = [ctx] + list(native_sides)
        for c_ in watched:
            c_.profile(True)
        for i in range(6):                                    # three recordings per demod stream
            one_alone(i)
        fence()
        for c_ in watched:
            for name, (ms, cnt) in c_.profile_read().items():
                a0 = alone.get(name, (0.0, 0))
                alone[name] = (a0[0] + ms, a0[1] + cnt)
            c_.profile(False)
    elif args.overlap and not loop_wl:
        saved, args.overlap = args.overlap, 0
        if args.steps > 1:                                    # (not for the one-step carrier-loop workloads: 5-8 s per step)
            run_steps(1)                                      # the sequential path's own buffers and first launches
            fence()
        ctx.profile(True)
        run_steps(3 if args.steps > 1 else 1)                 # a 30 us kernel measured once is whatever that one launch happened to be
        fence()
        alone = ctx.profile_read()
        ctx.profile(False)
        args.overlap = saved
    close_pipes()
    if loop_wl:
        sides.remove(engine.front)
        sides.remove(engine.tail)
        sides.remove(engine.slicing)
        if engine.loop is not None:
            sides.remove(engine.loop)
        lb.close_engines()                                    # the engine's work buffers and bitmap sets (gigabytes) go back
        chains_ref[:] = []
        import gc
        gc.collect()
        for c_ in [ctx] + [pymodem_amd.Context.side(dev_index, 400 + i) for i in range(8)]:
            c_.drop_scratch()                                 # ... and the slicers' output blocks of thousands of streams
    per_rank_ms = None
    if use_dist:
        mine_t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device or "cpu")
        every = [torch.zeros_like(mine_t) for _ in range(world)]
        torch.distributed.all_gather(every, mine_t)
        per_rank_ms = [round(float(v.item()) / args.steps * 1e3, 4) for v in every]
        elapsed = max(float(v.item()) for v in every)           # the slowest rank sets the job's time

    # end to end including the recording's way into HBM, overlapped: K more steps with every recording uploaded one step ahead
    h2d_overlapped = None
    pinned_source = False
    if args.overlap >= 2 and not loop_wl:
        import ctypes
        from pymodem_amd._native import lib as _lib
        pinned_source = os.environ.get("BENCH_PIN_SOURCE", "1") != "0" and \
            _lib().pm_host_pin(ctx.handle, audio.ctypes.data_as(ctypes.c_void_p), audio.nbytes) == 0
        run_steps_uploading(args.warmup)
        fence()
        t_u = time.perf_counter()
        run_steps_uploading(args.steps)
        fence()
        h2d_overlapped = time.perf_counter() - t_u
        close_pipes()
        if use_dist:
            tu = torch.tensor([h2d_overlapped], dtype=torch.float64, device=coll_device or "cpu")
            torch.distributed.all_reduce(tu, op=torch.distributed.ReduceOp.MAX)
            h2d_overlapped = float(tu.item())
    if pinned_source:
        _lib().pm_host_unpin(audio.ctypes.data_as(ctypes.c_void_p))
    # the recording's way into HBM, outside the timed region (the boundary hands over a host buffer): pageable int16 -> device
    t_up = time.perf_counter()
    d_again = ctx.upload(audio)
    ctx.sync()
    h2d_ms = (time.perf_counter() - t_up) * 1e3
    del d_again

    dinfo = dist_info(use_dist)                   # (a collective: every rank)
    if rank == 0:
        total_samples = float(args.samples) * nchains * args.steps
        value = total_samples / elapsed / 1e6
        # Roofline: the dominant kernel of the FIR/correlator stage (the stage north_star prices against HBM).  Algorithmic bytes
        # and flops are what the library accounted for the very launches that were timed (pm_prof_work).  The slicer and the
        # carrier loops are dependent-latency bound: neither roofline says anything about them, so they are
        # listed with their time in `gpu_kernel_ms_per_step` and named in `dominant_by_time` when they lead.
        by_time = max(prof, key=lambda k: prof[k][0])
        stage = [k for k in ("fir_i16", "fir_f64", "afsk_correlate", "signs") if prof[k][1]]
        dom = max(stage, key=lambda k: prof[k][0])
        dom_ms, dom_n = prof[dom]
        dom_bytes, dom_flops = work[dom]
        avg_ms = dom_ms / max(dom_n, 1)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        tflops = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic = pmc_traffic(args, dom)
        alone_ms = alone[dom][0] / max(alone[dom][1], 1) if alone.get(dom, (0, 0))[1] else None
        per_launch_bytes, per_launch_flops = dom_bytes / max(dom_n, 1), dom_flops / max(dom_n, 1)
        out = {
            "metric": "Msamples/s through demod_chain", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            # the native executor's certified sweeps: band-pass and low-pass sums as exact int8 digit products (matrix pipe) under a proven
            # bound, recombined in f64; everything else (sliding sums, combine, exact recomputation, every other executor) in f64
            "dtype": "f64+i8" if native_exec[0] else "f64", "data": "synthetic",
            "value_with_h2d_overlapped": None if h2d_overlapped is None else round(float(args.samples) * nchains * args.steps / h2d_overlapped / 1e6, 3),
            "steady_state_ms_per_step": steady.get("ms_per_step"),
            "steady_state_value": None if not steady else round(float(args.samples) * nchains / (steady["ms_per_step"] * 1e-3) / 1e6, 3),
            "ms_per_step_by_rank": per_rank_ms,
            "config": {"workload": f"{args.workload}: {desc}" + extra_chains_note(args, nchains), "chains_per_gpu": (cpg if args.scaling == "weak" else round(nchains / world, 3)), "chains_total": nchains,
                       "samples_per_recording": args.samples, "sample_rate": args.rate,
                       "buffer": BUFFER_DESC[args.buffer] + f", resident in HBM; successive recordings read DISTINCT copies ({len(ring)} of them in rotation)",
                       "parallelism": (f"chains sharded {cpg}/GPU x {world} GPU, packet gather to rank 0" if args.scaling == "weak" else
                                       f"the config's {nchains} chains divided over {world} GPU (contiguous blocks), packet gather to rank 0"),
                       "dist": dinfo,
                       "overlap": ("carrier-loop batch engine: every recording of a run in flight at once" if loop_wl else
                                   {0: "none", 1: "host half of step k behind GPU half of step k+1",
                                    2: "3-stage pipeline: demod(k+1) | slice(k) on a side stream | host(k-1)"}[min(args.overlap, 2)]),
                       "executor": ("native: pm_pipe_* per rank (one library call per recording; slicer batches and LFSR + codec on the library's own "
                                    "threads), the rows to dist.Exchanger, de-dup on rank 0" if native_exec[0] and exchanging else
                                    "native: pm_pipe_* (one library call per recording; slicer batches, LFSR + codec and de-dup on the library's "
                                    "own threads)" if native_exec[0] else
                                    None if loop_wl or args.overlap < 2 else "python: chain_execute.RecordingPipeline sequences the library's stage calls"),
                       "loop_batch": None if loop_info is None else dict(loop_info, phase_ms_per_step={k2: round(v / args.steps * 1e3, 3) for k2, v in loop_phase_s.items()}),
                       "scaling_note": ("carrier-loop workload: a GPU's time per run is one recording's worth of sequential loop steps however "
                                        "many loops run beside each other, so per-GPU throughput is set by recordings x chains in flight "
                                        "(--loop-batch x chains per GPU); sharding 8 chains per GPU over N GPUs scales by construction (no "
                                        "data-path collective; one packet gather per recording)" if loop_wl else None)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "binding_resource": valu_issue(args, dom, alone_ms, avg_ms),
                         "traffic_source": None if traffic is None else "committed rocprofv3 PMC passes of this workload (profiles/*_pmc.json: FETCH_SIZE "
                                           "and WRITE_SIZE in separate runs, x1024, FETCH doubled per the gfx950 note), per launch of this kernel class; "
                                           "counters cannot be read from inside the process",
                         "avg_kernel_ms": round(avg_ms, 5), "launches": dom_n, "algorithmic_bytes_per_launch": round(per_launch_bytes),
                         "in_flight": None if dom not in spans else {
                             "launches_in_flight_on_average": round(spans[dom]["sum_ms"] / max(spans[dom]["busy_ms"], 1e-9), 3),
                             "class_busy_ms_per_step": round(spans[dom]["busy_ms"] / args.steps, 5),
                             "achieved_over_busy_time": round(dom_bytes / (spans[dom]["busy_ms"] * 1e-3) / 1e9, 2),
                             "frac_over_busy_time": round(dom_bytes / (spans[dom]["busy_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "tflops_f64_over_busy_time": round(dom_flops / (spans[dom]["busy_ms"] * 1e-3) / 1e12, 3),
                             "note": "the executor runs this class on several streams at once (recordings take turns on the demod streams: two in the "
                                     "native executor): a launch that shares the GPU with others of its kind takes longer, so `achieved` (bytes of a launch / "
                                     "its own duration, what a kernel trace shows) falls as throughput rises.  Over-busy-time figures divide the "
                                     "class's bytes and flops by the time during which at least one of its launches was running (union of the "
                                     "HIP-event intervals of all streams)"},
                         "stage": "FIR/correlator", "dominant_by_time": by_time,
                         "alone": None if alone_ms is None else {
                             "avg_kernel_ms": round(alone_ms, 5), "achieved": round(per_launch_bytes / (alone_ms * 1e-3) / 1e9, 2),
                             "frac": round(per_launch_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "note": ("the same kernels through the same executor with ONE recording in flight (submitted, its result taken, then "
                                      "the next): six extra recordings after the timed region, no other stream on the GPU while a demod stage runs"
                                      if native_exec[0] and args.overlap >= 2 and not exchanging and args.steps > 1 else
                                      "same kernel class in three extra sequential steps after the timed region, no other stream on the GPU")},
                         "note": "achieved = algorithmic bytes of the timed launches / their HIP-event time inside the timed region (where "
                                 "the slicer streams share the CUs).  Above ~40 taps a FIR is bound by the vector-f64 FMA pipe, not by "
                                 "HBM: see roofline_fp64 for that fraction (DESIGN.md 4.1-4.2).  For the AFSK workloads the class fir_f64 is "
                                 "the fused certified kernel (sliding correlator sums + two 100-tap low-passes + combine for a whole gain "
                                 "sweep): it reads the band-passed stream once and writes sign bits only, so its algorithmic bytes are 8 B "
                                 "per sample for work that is 32 B per sample AND CHAIN stage by stage (SURVEY 8d) -- a low HBM fraction "
                                 "here means little traffic, not idle hardware; its HBM-bound parts measured on their own: sliding sums "
                                 "61 % and 8-tap FIR 71-79 % of peak (DESIGN.md 4.2c, 6).  Native executor, round 3: the band-pass (class fir_i16) "
                                 "and the fused kernel's low-passes run as int8 digit products on the matrix pipe (pm_bpf8.hip, "
                                 "afsk_slide_lpf8_kernel); bytes per launch are unchanged.  Round 5: the WHOLE AFSK stage of a chain group is one "
                                 "launch per recording (afsk_fused8_kernel: band-pass, both sweeps, exact recomputation) that reads the int16 "
                                 "recording once and writes one bit per sample and chain -- SURVEY 8(d)'s fused bound of 2 B per sample: "
                                 "`algorithmic_bytes_per_launch` is that (86 MB; it was 245 MB per sweep launch while the band-passed stream still "
                                 "went through memory), so `frac` FELL while the stage got faster: the kernel is bound by vector instruction "
                                 "issue (roofline_fp64, DESIGN.md 4.2), and `reference_dataflow` prices the same launch at the reference's own "
                                 "stage-by-stage traffic"},
            "reference_dataflow": None if args.workload != "afsk_1200_super_opt" else {
                "bytes_per_sample_and_chain": 50, "bytes_per_recording": round(50.0 * args.samples * len(my)),
                "equivalent_GBps": round(50.0 * args.samples * len(my) / max(dom_ms / max(dom_n, 1) * (1 if not prof["fir_i16"][1] else 2) + (prof["fir_i16"][0] / max(prof["fir_i16"][1], 1)), 1e-9) / 1e6, 1),
                "note": "SURVEY 8(d): the reference's AFSK chain moves 50 B per sample and chain stage by stage (BPF 10 + correlators 16 + LPF 16 + slicer "
                        "read 8); this is that traffic / the time of this rank's demod launches per recording -- an EQUIVALENT rate (it may exceed the "
                        "HBM peak: the intermediates it counts never exist here), beside `roofline`, which counts what does cross HBM"},
            "roofline_fp64": {"bound": "valu_f64", "kernel": dom, "achieved": round(tflops, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(tflops / FP64_PEAK_TFLOPS, 5), "algorithmic_flops_per_launch": round(per_launch_flops),
                              "alone_frac": None if alone_ms is None else round(per_launch_flops / (alone_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5),
                              "sustained_peak_measured": FP64_SUSTAINED_TFLOPS,
                              "alone_frac_of_sustained": None if alone_ms is None else round(per_launch_flops / (alone_ms * 1e-3) / 1e12 / FP64_SUSTAINED_TFLOPS, 5),
                              "note": "2 flops per fma of the FIR sums (epilogue sqrt / sign tests not counted) / the same HIP-event time; "
                                      "sustained_peak_measured is the rate a pure v_fma_f64 register loop holds on this chip (clock under f64 load)"
                                      + ("; NATIVE EXECUTOR: the low-pass and band-pass sums counted here are computed as int8 digit products on "
                                         "the matrix pipe (8 resp. 8 int8 digit products per tap and sample -- roofline_matrix.digit_pairs -- exact, recombined in f64): `achieved` is "
                                         "the rate of the f64 sums they stand for, an EQUIVALENT rate -- it may exceed what the vector pipe "
                                         "sustains and is not a utilisation of it" if native_exec[0] else "")},
            "roofline_matrix": matrix_roofline(native_exec[0], args, modems, my, prof),
            "roofline_by_class": {k: {"avg_kernel_ms": round(prof[k][0] / prof[k][1], 5), "launches": prof[k][1],
                                      "algorithmic_bytes_per_launch": round(work[k][0] / prof[k][1]),
                                      "achieved_GBps": round(work[k][0] / (prof[k][0] * 1e-3) / 1e9, 1),
                                      "frac_hbm": round(work[k][0] / (prof[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                      "tflops_f64": round(work[k][1] / (prof[k][0] * 1e-3) / 1e12, 3),
                                      "frac_f64_sustained": round(work[k][1] / (prof[k][0] * 1e-3) / 1e12 / FP64_SUSTAINED_TFLOPS, 5)}
                                  for k in stage if prof[k][0] > 0 and work[k][0] > 0},
            "gpu_kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]},
            "h2d": {"ms": round(h2d_ms, 3), "bytes": int(audio.nbytes),
                    "value_with_h2d": round(float(args.samples) * nchains / (elapsed / args.steps + h2d_ms * 1e-3) / 1e6, 3),
                    "value_with_h2d_overlapped": None if h2d_overlapped is None else round(float(args.samples) * nchains * args.steps / h2d_overlapped / 1e6, 3),
                    "overlapped_source_page_locked": bool(pinned_source),
                    "note": "value_with_h2d: one upload of the recording from pageable host memory, measured after the timed region and added "
                            "to every step un-overlapped.  value_with_h2d_overlapped (also a top-level field): a second timed run of the same K "
                            "steps in which every step's recording starts in page-locked host memory and is copied to HBM one step ahead on a "
                            "copy stream (RecordingPipeline.prefetch).  `value` itself has the recording resident in HBM"},
            "pipeline_stage_ms_per_step": stage_ms or None,
            "host_cpu_ms_per_step": round(host_cpu_ms, 3), "host_cores_visible": len(os.sched_getaffinity(0)),
            "slicer": chains_ref[0][2].last_stats if chains_ref else None,
            "packets": {"unique_good": result.CountGood() if result is not None else None,
                        "bad": result.CountBad() if result is not None else None},
        }
        return out
    return None


def extra_chains_note(args, nchains):
    """Names the chains a weak-scaling run adds beyond the config's own (they are not in the bundled file)."""
    own = {"afsk_1200_super_opt": 8, "fsk_9600": 3, "bpsk_300": 1, "qpsk_2400": 64}[args.workload]
    if nchains <= own:
        return ""
    factory = WORKLOADS[args.workload][0]
    names = [factory(c)["object_name"] for c in range(own, nchains)]
    shown = ", ".join(names[:4]) + (f", ... ({len(names)} in all)" if len(names) > 4 else "")
    return f" + {len(names)} chains beyond the config's {own} (same modem, the sweep continued: {shown})"


def dist_info(use_dist):
    """What torch.distributed saw (the exchange's world), or None for a plain one-process run."""
    if not use_dist:
        return None
    import torch.distributed as dist
    seen = dist.get_world_size()
    try:                                          # every rank says which GPU it drives: what the collectives really span
        import torch
        mine = [torch.cuda.current_device() if torch.cuda.is_available() else -1, int(os.environ.get("LOCAL_RANK", "0"))]
        every = [None] * seen
        dist.all_gather_object(every, mine)
    except Exception:                             # noqa: BLE001
        every = None
    return {"world_size": seen, "backend": dist.get_backend(), "rank0_of": seen,
            "ranks_seen_by_rccl": seen if dist.get_backend() == "nccl" else 0, "device_and_local_rank_by_rank": every}


def valu_issue(args, kernel_class, alone_ms, pipeline_ms):
    """What does bind the headline's dominant kernel, from the committed SQ counter passes (profiles/*_pipeline_kernel_counters.json,
    tools/collect_pipeline_counters.sh): vector instructions per launch x 4 cycles (a wave64 instruction occupies its SIMD's 16-lane
    vector pipe for four) over the SIMD cycles of the launch at the nominal 2.4 GHz on 1024 SIMDs.  None when no profile matches."""
    import glob
    if args.workload != "afsk_1200_super_opt" or kernel_class != "fir_f64" or args.samples != 28_800_000:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pipeline_kernel_counters.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:                                         # noqa: BLE001
            continue
        k = [n for n in d if n.startswith("afsk_fused8_kernel")]
        if not k:
            continue
        c = d[k[0]]
        valu, mfma_busy = c["SQ_INSTS_VALU"]["avg"], c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"]
        simd_cycles = lambda ms: ms * 1e-3 * 2.4e9 * 1024
        out = {"resource": "vector instruction issue", "kernel": k[0], "vector_instructions_per_launch": round(valu),
               "matrix_busy_cycles_per_launch": round(mfma_busy), "source": os.path.basename(path),
               "note": "vector instructions x 4 cycles / (launch time x 2.4 GHz x 1024 SIMDs): the share of the chip's SIMD cycles the launch spends "
                       "issuing vector instructions -- the resource this kernel is bound by; the HBM roofline above counts its 86 MB of traffic "
                       "and stays at a few percent whatever the kernel does"}
        if alone_ms:
            out["simd_cycle_frac_alone"] = round(4.0 * valu / simd_cycles(alone_ms), 4)
            out["matrix_pipe_frac_alone"] = round(mfma_busy / simd_cycles(alone_ms), 4)
        if pipeline_ms:
            out["simd_cycle_frac_in_pipeline_per_launch"] = round(4.0 * valu / simd_cycles(pipeline_ms), 4)
        return out
    return None


def pmc_traffic(args, kernel_class):
    """HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes of the same command (profiles/*_pmc.json:
    FETCH_SIZE and WRITE_SIZE in separate passes, x1024, FETCH doubled per the gfx950 note).  PMC counters cannot be read from
    inside the process, so this is the offline measurement; null when no profile matches the workload and size."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") != args.workload or d.get("samples") != args.samples:
            continue
        v = d.get("classes", {}).get(kernel_class)
        if v and v.get("traffic_bytes_per_launch"):
            return v["traffic_bytes_per_launch"]
    return None


_CPU_AUDIO = None


def _cpu_worker(job):
    """One process = one core: whole chain passes of its chain over the shared sample until the time budget is spent."""
    rate, line, seconds, max_passes = job
    from oracle import oracle as O
    t0 = time.perf_counter()
    done = 0
    while True:
        O.run_chain(O.build_chain(rate, line), _CPU_AUDIO, canon=False)
        done += 1
        if time.perf_counter() - t0 > seconds or done >= max_passes:
            break
    return done, time.perf_counter() - t0


def cpu_baseline(args, lines, seconds=12.0, max_passes=16):
    """The oracle (CPU restatement of the reference: numpy.convolve FIRs + C loops + Python codecs) on a bounded sample of the same
    workload: one process per chain, one core each, forked BEFORE anything touches the GPU.  A reported baseline, not
    the optimisation target."""
    global _CPU_AUDIO
    import multiprocessing as mp
    n = args.cpu_sample or min(args.samples, 9_600_000)          # 200 s of audio per chain keeps numpy's buffers modest
    _CPU_AUDIO = make_buffer(args)[:n]
    procs = max(1, min(len(lines), (os.cpu_count() or 2) // 2, 8))
    jobs = [(args.rate, lines[k % len(lines)], seconds, max_passes) for k in range(procs)]
    t0 = time.perf_counter()
    if procs == 1:
        results = [_cpu_worker(jobs[0])]
    else:
        with mp.get_context("fork").Pool(procs) as pool:        # children share the sample copy-on-write
            results = pool.map(_cpu_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in results)
    busy = max(r[1] for r in results)
    _CPU_AUDIO = None
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(n * done / busy / 1e6, 3), "unit": "Msamples/s", "cores": procs, "kind": "port", "cpu_model": model,
            "host_cores_available": os.cpu_count(), "per_core": round(n * done / busy / 1e6 / procs, 3),
            "sample": f"{done} chain passes ({procs} processes, one chain each of the workload's {len(lines)}) x the first {n} samples of the "
                      f"same buffer, oracle = numpy.convolve FIRs + C loops + Python codecs, {busy:.1f} s busy / {wall:.1f} s wall"}


if __name__ == "__main__":
    main()
