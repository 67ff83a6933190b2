#!/usr/bin/env python3
"""bench.py -- MLL+grad iterations/sec of the exact Projected-LMC training step
(BASELINE.json metric).

Workload (configs[2], "C3"): ProjectedGPModel + ProjectedLMCmll, n=8192 points, d=8, p=16 tasks,
q=8 latents, Matern-5/2, fp32, variant PLMC_fast (experiments.py:211-215), synthetic data with
the structure of experiments.py:137-167.  One step = the body of experiments.py:264-273:
zero_grad -> model(X) -> -mll -> backward -> (all-reduce of grads when sharded) -> AdamW step.

N GPUs: the q latent GPs are sharded across ranks (latent i -> rank i mod N), total work fixed
(strong scaling); the only collective is one small fused all-reduce per step.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # N > 1 outside torch.distributed.run: starts the N ranks itself (self_launch)
"""
import argparse
import json
import math
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6, "bf16": 2500.0}     # MI355X_MICROARCH.md chip table (dense matrix peaks)
SPLIT_PRODUCTS = {2: 3, 3: 6}          # 16-bit plane products per fp32 product in the split engines (csrc/bf3_engine.hpp)
ARITHMETIC = {
    2: "fp32 storage, results and accumulation; the bulk products (tail / head updates and group panels of the sweep, K^-1 = W^T W) "
       "on v_mfma_f32_16x16x32_f16 from operands split into two fp16 planes (s x = h0 + 2^-11 h1, 22 bits; s = a power of two from "
       "bounds by the largest diagonal entry and the noise), three plane products into two fp32 accumulator levels (error vs fp64 "
       "0.3-0.45 x the fp32 MFMA chain's, profiles/r03_split_numerics.txt); chain, next-group panel columns, look-ahead updates and "
       "augmented columns on v_mfma_f32_16x16x4_f32",
    3: "as the default, but the bulk products on v_mfma_f32_16x16x32_bf16 from three bf16 planes (x = hi + mid + lo exactly), six plane "
       "products into two fp32 accumulator levels (PLMC_SPLIT=3)",
    0: "fp32 storage and arithmetic: every product on v_mfma_f32_16x16x4_f32 (PLMC_SPLIT=0)"}


def make_data(n, d, p, q, seed=0, dtype=torch.float32):
    """X ~ U(-1,1)^(n x d); Y = G H_true (1 - mu_noise) + structured + unstructured noise, mirroring
    experiments.py:137-167.  Latent draws G use random Fourier features of the Matern-5/2 spectral
    density (multivariate t, 5 dof) so that no n x n host factorisation is needed at n = 8192."""
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    lsc = torch.linspace(0.1, 0.5 * math.sqrt(d), q, dtype=torch.float64)
    nf = 1024
    G = torch.empty(n, q, dtype=torch.float64)
    for i in range(q):
        z = torch.randn(nf, d, generator=g, dtype=torch.float64)
        u = (torch.randn(nf, 5, generator=g, dtype=torch.float64) ** 2).sum(1, keepdim=True)   # chi^2_5
        om = z / lsc[i] * torch.sqrt(5.0 / u)
        ph = 2 * math.pi * torch.rand(nf, generator=g, dtype=torch.float64)
        w = torch.randn(nf, generator=g, dtype=torch.float64)
        G[:, i] = math.sqrt(2.0 / nf) * (torch.cos(X @ om.T + ph) @ w)
    mu_noise, mu_str = 0.1, 0.9
    H_true = torch.randn(q, p, generator=g, dtype=torch.float64)
    Y_sig = G @ H_true * (1 - mu_noise)
    H_hid = torch.randn(q, p, generator=g, dtype=torch.float64)
    Y_com = torch.randn(n, q, generator=g, dtype=torch.float64) @ H_hid * mu_str
    lev = torch.rand(p, generator=g, dtype=torch.float64) + 0.1
    Y_spec = torch.randn(n, p, generator=g, dtype=torch.float64) * torch.sqrt(lev)[None, :] * (1 - mu_str)
    Y = Y_sig + (Y_com + Y_spec) * mu_noise
    return X.to(dtype), Y.to(dtype)


def build_key():
    """sha256 over the library sources: PMC traffic numbers under profiles/ are only valid for the build they were
    measured on (tools/pmc_aggregate.py stores the same key)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "projected-lmc_amd", "csrc")
    for f in sorted(os.listdir(csrc)) + [os.path.join("..", "..", "include", "plmc.h")]:
        path = os.path.join(csrc, f)
        if os.path.isfile(path) and f.endswith((".hip", ".hpp", ".h")):
            h.update(f.encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(kname):
    """HBM bytes per launch of `kname` from the newest profiles/rNN_pmc_traffic.json measured on THIS build, else None."""
    import glob
    key = build_key()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except Exception:
            continue
        if pmc.get("build_key") == key and kname in pmc.get("kernels", {}):
            return pmc["kernels"][kname]["hbm_bytes_corrected"], os.path.relpath(path, ROOT)
    return None, None


def committed_counters(kname):
    """MfmaUtil (%) and L2 hit rate of `kname` from the newest profiles/rNN_pmc_mfma_l2.json measured on THIS build
    (tools/collect_pmc_extra.sh), else {}."""
    import glob
    key = build_key()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma_l2.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except Exception:
            continue
        k = pmc.get("kernels", {}).get(kname)
        if pmc.get("build_key") == key and k:
            return {"mfma_util_pct": k.get("MfmaUtil"), "l2_hit_rate": k.get("l2_hit_rate"), "counters_source": os.path.relpath(path, ROOT)}
    return {}


def live_traffic(kname, latents):
    """Two rocprofv3 --pmc child runs of this script (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only), aggregated
    as tools/pmc_aggregate.py does (hbm bytes = 2 * FETCH + WRITE on gfx950, MI355X_MICROARCH.md).  Returns bytes per launch or None."""
    import csv, glob, shutil, subprocess, tempfile
    tot = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="plmc_pmc_")
        try:
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", tmp, "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--latents", str(latents), "--no-cpu-baseline", "--no-prof"]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=240, check=True)
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            vals = [float(r["Counter_Value"]) * 1024.0 for r in csv.DictReader(open(files[0]))
                    if r.get("Counter_Name") == ctr and kname in r["Kernel_Name"].replace("plmc::", "")]
            tot[ctr] = sum(vals) / max(1, len(vals))
        except Exception:
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return 2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]


def host_cores():
    """Cores this process may actually use: min(affinity, cgroup CPU quota).  (On the GPU box
    os.cpu_count() reports the whole host while the container is limited by cpu.max.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def oracle_latent(X, ell, noise, ytil):
    """fp64 oracle of one latent at full size: (logp, gradient vector over lengthscales, noise and every y entry)."""
    from oracle import cpu_step
    lp, ge, gn, gy = cpu_step.latent_step("matern", X.double(), ell.double(), noise.double(), ytil.double(), nu=2.5)
    return float(lp), torch.cat([ge.reshape(-1), gn.reshape(-1), gy.reshape(-1)])


def cpu_baseline(X, Y, P, n_latents):
    """The oracle ("port": plain torch-CPU restatement, all host threads) timed on ONE full training step of the same
    workload -- the loop body of experiments.py:264-273: projection, the q latent exact-GP MLL+gradient evaluations at full n
    (fp32 dense Cholesky + cholesky_inverse, analytic gradient), projection terms, backward, AdamW step
    (oracle/cpu_step.py projected_step) -- after one untimed latent evaluation (thread pools, page-in).  iters/sec = 1 / t_step."""
    from oracle import cpu_step
    from oracle import projected as pj
    n = X.shape[0]
    with torch.no_grad():
        cpu_step.latent_step(P["kind"], X, pj.lengthscale(P)[0], pj.projected_noise(P)[0], pj.project_data(P, Y)[0], nu=P["nu"])
    t0 = time.time()
    loss, _ = cpu_step.projected_step(P, X, Y)
    t = time.time() - t0
    return dict(value=1.0 / t, unit="iters/sec", cores=torch.get_num_threads(), kind="port",
                sample="one full training step (projection + %d latent exact-GP MLL+gradient evaluations at n=%d, fp32 dense "
                       "Cholesky + inverse with the analytic gradient + projection terms + backward + AdamW; torch CPU) after one "
                       "untimed latent evaluation: %.1f s; loss %.6f" % (n_latents, n, t, loss))


def self_launch(n_ranks):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: run the N ranks as CHILD processes under
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1, a free port) and pass on their exit code.  Rank 0's
    JSON line goes straight to the inherited stdout.  This process never touches the GPU (nothing below `import torch`
    has run yet) and never replaces itself: it waits for the children (a process that has initialised HIP must not exec)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n_ranks)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--tasks", type=int, default=16)
    ap.add_argument("--latents", type=int, default=8)
    ap.add_argument("--variant", default="PLMC_fast", choices=["PLMC_fast", "PLMC"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-options", action="store_true", help="skip the runs with PLMC_SPLIT=0 (fp32 MFMA everywhere) and PLMC_SPLIT=3 (three bf16 planes) that are reported beside the headline")
    ap.add_argument("--pmc", action="store_true", help="measure roofline.traffic live (two rocprofv3 --pmc child runs) "
                                                      "when no profile of this build is committed")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        # the CPU legs (cpu_baseline, accuracy oracle) use every host core this process may use; set ONCE, before any CPU work of
        # the process (changing the thread count after OpenMP work has run is what not to do: tests/test_gpu_metric_shape.py)
        torch.set_num_threads(host_cores())
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch `python bench.py --gpus N` (it starts the ranks itself) or "
                         "torch.distributed.run --nproc-per-node N bench.py --gpus N" % (args.gpus, world))
    # PLMC_DIST_BACKEND=gloo is a rehearsal aid: several ranks sharing ONE GPU (RCCL refuses duplicate
    # devices) still exercise the sharded step end to end; the measured runs use "nccl" (= RCCL).
    # PLMC_BENCH_DEVICE=cpu: launch rehearsal on a box without a GPU (rendezvous, sharding, all-reduce, JSON line).  The
    # product has no CPU engine -- the first log-likelihood call raises on CPU tensors -- so this only gets anywhere inside
    # tests/test_bench_launch.py, whose child processes replace that call by a test stand-in.
    backend = os.environ.get("PLMC_DIST_BACKEND", "nccl")
    on_gpu = os.environ.get("PLMC_BENCH_DEVICE", "cuda") != "cpu"
    if on_gpu:
        dev_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    else:
        dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import projectedlmc as plmc
    from projectedlmc import _hip, parallel

    n, d, p, q = args.n, args.d, args.tasks, args.latents
    dt = torch.float32
    X, Y = make_data(n, d, p, q, seed=0, dtype=dt)
    kw = dict(BDN=True, diagonal_B=True, scalar_B=True) if args.variant == "PLMC_fast" else dict(BDN=False)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = plmc.ProjectedGPModel(X, Y, p, q, proj_likelihood=None, mean_type=plmc.ZeroMean,
                                      kernel_type=plmc.MaternKernel, init_lmc_coeffs=True,
                                      latent_shard=(rank, world) if world > 1 else None, **kw)
    # initial state for the CPU baseline / log-lik check (before moving to the device)
    cpu_params = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import bridge
        cpu_params = bridge.oracle_params(model, dtype=dt)                 # raw parameters of the initial model, oracle form
    with torch.no_grad():
        cpu_state = (model.covar_module.lengthscale.reshape(q, d).clone(), model.projected_noise().clone(),
                     model.project_data(Y).clone())
    model = model.to(dev)
    Xd, Yd = X.to(dev), Y.to(dev)
    model.train()
    model.likelihood.train()
    mll = plmc.ProjectedLMCmll(model.likelihood, model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    params = list(model.parameters())

    def step():
        opt.zero_grad()
        out = model(Xd)
        loss = -mll(out, Yd)
        loss.backward()
        gl = parallel.sync_loss_and_grads(loss, params)
        opt.step()
        return gl

    def fence():
        if on_gpu:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(dev)

    # Accuracy checkpoints (untimed): the HIP log-prob and its gradient w.r.t. EVERY parameter of a latent GP
    # (lengthscales, noise, all n projected targets) against the fp64 oracle at full size -- latent 0 at the initial
    # parameters, latents 0 and q-1 after 5 optimiser steps.  The mixing-matrix / noise-model parameters
    # sit behind d logp / d y~ through the same torch autograd on both sides.
    from projectedlmc import _engine
    do_checks = rank == 0 and world == 1 and not args.no_cpu_baseline
    checkpoints = []                                    # (label, latent, ell, noise, ytil, lp_gpu, grad_gpu)

    def gpu_checkpoint(label, state, latents):
        for j in latents:
            e = state[0][j:j + 1].to(dev).requires_grad_()
            z = state[1][j:j + 1].to(dev).requires_grad_()
            yt = state[2][j:j + 1].to(dev).requires_grad_()
            lp = _engine.exact_latent_log_prob("matern52", Xd, e, None, z, yt)
            lp.sum().backward()
            g = torch.cat([e.grad.reshape(-1), z.grad.reshape(-1), yt.grad.reshape(-1)]).double().cpu()
            checkpoints.append((label, j, state[0][j].cpu(), state[1][j].cpu(), state[2][j].cpu(), float(lp), g))

    if do_checks:
        gpu_checkpoint("initial parameters", cpu_state, [0])

    def note(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    note("warm-up (%d steps)" % args.warmup)
    first_loss = None
    st5 = None
    n_check = min(5, args.warmup)
    for i in range(args.warmup):
        l0 = step()
        if first_loss is None:
            first_loss = float(l0)
        if do_checks and i + 1 == n_check:
            with torch.no_grad():
                st5 = (model.covar_module.lengthscale.reshape(q, d).clone(), model.projected_noise().clone(),
                       model.project_data(Yd).clone())
            gpu_checkpoint("after %d optimiser steps" % n_check, st5, sorted({0, q - 1}))
            model.train()
    # Inside the timed region only the two MFMA-heavy kernel classes (and the whole sweep) are bracketed by
    # HIP events: a bracket costs ~10 us of stream time, which on the ~200 short chain kernels of a step
    # would distort the very step time being measured.  The full per-kernel table comes from extra untimed
    # steps afterwards.
    prof = not args.no_prof
    HEAVY = ("k_trail", "k_trail_head", "k_kinv_grad", "sweep_total")
    if prof:
        _hip.prof_enable(HEAVY)
        _hip.prof_collect()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step()
    fence()
    elapsed = time.perf_counter() - t0
    stats = _hip.prof_collect() if prof else {}
    table, table_steps, iso = {}, 3, {}
    if prof:
        _hip.prof_enable(True)
        for i in range(table_steps):
            step()
        fence()
        table = _hip.prof_collect()
        # the same kernels with the look-ahead off (one stream, nothing overlaps): per-kernel rates undiluted
        # by concurrent launches -- reported beside the in-situ roofline, never as the step time
        with _hip.knob("PLMC_SERIAL", "1"):
            for i in range(2):
                step()
            fence()
            iso = _hip.prof_collect()
        _hip.prof_enable(False)
    # The same steps with the other arithmetics of the bulk products -- PLMC_SPLIT=0: every product on the fp32 matrix
    # instruction; PLMC_SPLIT=3: three bf16 planes -- reported BESIDE the headline with their own accuracy checks at the same
    # checkpoint states (a split engine is the default only while its errors are at or below the fp32 path's).
    split = int(os.environ.get("PLMC_SPLIT", "2"))
    split = split if split in (0, 3) else 2
    options = {}
    if world == 1 and not args.no_options:
        for alt in (0, 3):
            if alt == split:
                continue
            tag = " [split %d]" % alt
            try:
                with _hip.knob("PLMC_SPLIT", str(alt)):
                    for i in range(2):
                        step()
                    fence()
                    t0 = time.perf_counter()
                    for i in range(args.steps):
                        step()
                    fence()
                    el = time.perf_counter() - t0
                    options[alt] = {"ms_per_step": 1e3 * el / args.steps, "iters_per_sec": args.steps / el}
                    if do_checks and st5 is not None:
                        gpu_checkpoint("after %d optimiser steps%s" % (n_check, tag), st5, sorted({0, q - 1}))
                        model.train()
            except Exception as exc:                            # a comparison run must never cost the headline line
                options[alt] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                checkpoints[:] = [c for c in checkpoints if not c[0].endswith(tag)]
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)

    if rank == 0:
        its = args.steps / elapsed
        res = {
            "metric": "MLL+grad iters/sec, n=8192 16-task LMC", "value": its, "unit": "iters/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "arithmetic": ARITHMETIC[split],
            "data": "synthetic",
            "config": {"workload": "C3: ProjectedGPModel+ProjectedLMCmll exact training step (%s), n=%d d=%d p=%d q=%d "
                                   "Matern-5/2 fp32; latents sharded q/N per GPU" % (args.variant, n, d, p, q),
                       "n_points": n, "n_dim": d, "n_tasks": p, "n_latents": q, "parallelism": "latent-shard x%d" % world},
            "final_loss": float(last),
        }
        if world > 1:
            res["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                  "collective": "one fused all-reduce (sum) of [loss share | all parameter gradients] per step"}
        # ---- roofline of the dominant kernel (largest share of HIP-event time in the timed region)
        if stats:
            sweep = stats.pop("sweep_total", None)          # wall time of plmc_potrf on the main stream
            mf = {k: v for k, v in stats.items() if v["flops"] > 0}
            dom = max(mf, key=lambda k: mf[k]["ms"])
            s = mf[dom]
            ach = s["flops"] / (s["ms"] * 1e-3) / 1e12
            # the three bracketed classes are exactly the launches the split engine carries: their peak is the dense bf16 peak
            # divided by the six plane products one fp32 product costs (fp32-equivalent TFLOP/s); with PLMC_SPLIT=0 the fp32 peak
            peak = MFMA_PEAK_TFLOPS["bf16"] / SPLIT_PRODUCTS[split] if split else MFMA_PEAK_TFLOPS["f32"]
            try:                                                  # bare instruction stream on this device, now (f16 runs at the bf16 rate)
                peak_meas = (_hip.mfma_rate("bf16", dev) / SPLIT_PRODUCTS[split]) if split else _hip.mfma_rate(torch.float32, dev)
            except Exception:
                peak_meas = None
            res["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "peak_note": ("fp32-equivalent: dense 16-bit MFMA peak %.0f / %d plane products" % (MFMA_PEAK_TFLOPS["bf16"], SPLIT_PRODUCTS[split]))
                               if split else "dense fp32 MFMA peak",
                               "frac": ach / peak, "traffic": None, "peak_measured": peak_meas,
                               "frac_of_measured_peak": (ach / peak_meas) if peak_meas else None,
                               "avg_launch_ms": s["ms"] / s["launches"], "launches": s["launches"],
                               "flops_per_launch": s["flops"] / s["launches"]}
            if dom in iso and iso[dom]["ms"] > 0:
                ia = iso[dom]["flops"] / (iso[dom]["ms"] * 1e-3) / 1e12
                res["roofline"]["isolated"] = {"achieved": ia, "frac": ia / peak, "launches": iso[dom]["launches"],
                                               "note": "same kernel, look-ahead off (PLMC_SERIAL=1, 2 untimed steps): no "
                                                       "concurrent launches share the GPU; `achieved` above is in situ, "
                                                       "where the chain + head run beside this kernel"}
            # HBM bytes per launch from the PMC counters.  They cannot be read from inside the process: they come from
            # rocprofv3 --pmc passes of this same command -- the committed ones if they were measured on THIS build
            # (build key over the library sources), else (--pmc) two child runs now, else null.
            sname = {2: "SplitH2", 3: "SplitB3"}.get(split)      # as tools/pmc_aggregate.py writes the names (plmc:: stripped)
            kname = ({"k_trail": "k_update_bf3<%s, 0>" % sname, "k_trail_head": "k_update_bf3<%s, 3>" % sname,
                      "k_kinv_grad": "k_kinv_grad_bf3<%s, 8, false>" % sname} if split else
                     {"k_trail": "k_update<float, 0, 4>", "k_trail_head": "k_update<float, 3, 4>",
                      "k_kinv_grad": "k_kinv_grad<float, 8, false>"}).get(dom, dom + "<float>")
            if world == 1:
                tr, src = committed_traffic(kname)
                if tr is None and args.pmc:
                    tr, src = live_traffic(kname, q), "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs"
                res["roofline"]["traffic"] = tr
                res["roofline"].update(committed_counters(kname))       # isolated MfmaUtil / L2 hit rate of the same kernel, if profiled on this build
                res["roofline"]["traffic_note"] = (
                    "bytes/launch, 2*FETCH_SIZE + WRITE_SIZE (%s, build %s); algorithmic tile traffic/launch = %.3g"
                    % (src, build_key(), s["bytes"] / s["launches"]) if tr is not None else
                    "no PMC profile of build %s under profiles/ (run tools/collect_profiles.sh or bench.py --pmc)" % build_key())
            # per-kernel table: every class bracketed, from the untimed steps after the timed region
            table.pop("sweep_total", None)
            res["kernels"] = {k: {"ms_per_step": v["ms"] / table_steps, "launches_per_step": v["launches"] / table_steps,
                                  "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] > 0 and v["ms"] > 0 else None,
                                  "gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
                              for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]) if v["launches"] > 0}
            res["kernels_note"] = "from %d extra untimed steps with every kernel bracketed; roofline/cholesky_gemm are from the timed steps" % table_steps
            # whole-step fractions the north_star asks for (local latents only).  The blocked sweep
            # (k_diag + k_panel + k_trail) produces BOTH the Cholesky factor U and the inverse factor
            # W = U^-T: F_sweep = 2 q n^3/3 ("Cholesky-GEMM roofline"); F_step = q n^3 over the whole step.
            q_loc = len(range(rank, q, world))
            chol_ms = sweep["ms"] / args.steps
            f_sweep = 2.0 * q_loc * n ** 3 / 3
            p32 = MFMA_PEAK_TFLOPS["f32"]
            res["cholesky_gemm"] = {"what": "factor U and inverse factor W in one sweep, 2 q n^3 / 3 flop",
                                    "tflops": f_sweep / (chol_ms * 1e-3) / 1e12, "ms_per_step": chol_ms,
                                    "frac_of_fp32_mfma_peak": f_sweep / (chol_ms * 1e-3) / 1e12 / p32}
            res["step_dense"] = {"tflops": q_loc * n ** 3 / (elapsed / args.steps) / 1e12,
                                 "frac_of_fp32_mfma_peak": q_loc * n ** 3 / (elapsed / args.steps) / 1e12 / p32,
                                 "note": "q n^3 flop per step over the step time, against the fp32 matrix peak %.1f TF (the step may "
                                         "exceed it: its bulk products run on the bf16 matrix cores)" % p32}
        note("%.2f ms/step on %d GPU(s)" % (1e3 * elapsed / args.steps, world))
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle on %d host cores (one full step of the same workload) ..." % host_cores())
            cb = cpu_baseline(X, Y, cpu_params, q)
            res["cpu_baseline"] = cb
            res["speedup_vs_cpu"] = its / cb["value"]
            note("fp64 oracle at the %d accuracy checkpoints ..." % len(checkpoints))
            checks = []
            opt_checks, memo = {}, {}
            for label, j, e, z, yt, lp_gpu, g_gpu in checkpoints:
                key = (j, label.split(" [split")[0])
                if key not in memo:
                    memo[key] = oracle_latent(X, e, z, yt)
                lp_cpu, g_cpu = memo[key]
                alt = int(label.split("[split ")[1][:-1]) if "[split " in label else None
                (opt_checks.setdefault(alt, []) if alt is not None else checks).append(
                    {"where": "latent %d, %s" % (j, label), "logp_hip_f32": lp_gpu, "logp_oracle_f64": lp_cpu,
                     "loglik_rel_err": abs(lp_gpu - lp_cpu) / abs(lp_cpu),
                     "grad_rel_err": float((g_gpu - g_cpu).norm() / g_cpu.norm()),
                     "grad_max_abs_err_over_max": float((g_gpu - g_cpu).abs().max() / g_cpu.abs().max())})
            res["loglik_rel_err"] = max(c["loglik_rel_err"] for c in checks)
            res["grad_rel_err"] = max(c["grad_rel_err"] for c in checks)
            res["accuracy_checks"] = checks
            for alt, oc in opt_checks.items():
                if alt in options and "error" not in options[alt]:
                    options[alt]["loglik_rel_err"] = max(c["loglik_rel_err"] for c in oc)
                    options[alt]["grad_rel_err"] = max(c["grad_rel_err"] for c in oc)
                    options[alt]["accuracy_checks"] = oc
            res["accuracy_note"] = ("log N(y~; 0, K + s2 I) of a latent GP and its gradient w.r.t. every lengthscale, the noise and "
                                    "all n projected targets (d + 1 + n numbers), fp32 HIP vs fp64 CPU oracle at n = %d; max over the "
                                    "checkpoints in loglik_rel_err / grad_rel_err" % n)
        res["first_loss"] = first_loss
        for alt, opt in options.items():
            opt["what"] = ("NOT the headline: the same %d steps with PLMC_SPLIT=%d -- " % (args.steps, alt)) + ARITHMETIC[alt]
            res["fp32_mfma_option" if alt == 0 else "bf16x3_option"] = opt
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
