#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per second of the IndexTTS-1.5 hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run with one rank per GPU.  One "step" = one pass of the whole hot path over one batch
of synthetic utterances per GPU: conditioning (conformer+perceiver) + ECAPA once per
utterance, greedy AR decode with KV cache (all sentences of the utterance as one batch, fixed length,
eos suppressed), latent pass per sentence, BigVGAN, int16 waveform copied to the host (infer.py:208-212).
Inputs (prompt mel, weights) are resident in HBM before the timed region.  Utterances shard over ranks with
no collective on the data path (weights are replicated once by an RCCL broadcast before timing).

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (torch.distributed.run as a CHILD
process, before this process touches the GPU) and relays rank 0's JSON line.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "index-tts-ipex_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step (config 2: 1, config 3: 32)")
    ap.add_argument("--sentences", type=int, default=2, help="sentences per utterance (200-char zh = 2 x L105)")
    ap.add_argument("--text-tokens", type=int, default=105)
    ap.add_argument("--mel-tokens", type=int, default=480)
    ap.add_argument("--prompt-frames", type=int, default=511)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "fp32"],
                    help="engine storage dtype: bf16 (BASELINE config 2), f16 = IEEE half (the reference's is_fp16=True), fp32 (parity engine)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--micro", action="store_true", help="tiny config (plumbing check)")
    ap.add_argument("--gpt-fp8", action="store_true",
                    help="BASELINE config 5 storage: GPT projections as fp8 e4m3 + row scales for the decode GEMV (bf16 activations / KV)")
    ap.add_argument("--beams", type=int, default=1,
                    help="HF beam-sample with this many beams per sentence (the reference's default generate() mode is 3: top_k 30, "
                         "top_p 0.8, temperature 1.0) instead of greedy; decode rows = sentences x beams")
    ap.add_argument("--sample", action="store_true", help="HF sample() (top_k 30, top_p 0.8, temperature 1.0, one beam) instead of greedy")
    ap.add_argument("--no-graph", action="store_true", help="eager decode launches (for rocprofv3 --pmc passes)")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements of the default run (BASELINE config 3 / 4, product loop)")
    ap.add_argument("--eos", action="store_true",
                    help="decode with the product loop of Engine.generate: a status() read-back (stream sync + D2H of the unfinished "
                         "flags) every 16 steps, as with eos enabled; the stop token stays suppressed so the work is the same")
    return ap.parse_args()


def launch_ranks(a) -> int:
    """`--gpus N` outside torchrun: start N ranks as a child `torch.distributed.run` (this process has not initialised
    the GPU and never will), relay their output, return the child's exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def build_engine_dp(cfg, dtype, device, gpt_fp8=False, max_batch=128):
    """Rank 0 materialises + packs the synthetic checkpoint; the other ranks receive the packed arenas by one broadcast
    each (RCCL over xGMI), outside the timed region: itts_hip.dp.replicate_packed, the path tests/test_dp_gloo.py covers."""
    from itts_hip import dp
    from itts_hip import engine as ieng
    from itts_hip import pack, synth

    eng = ieng.Engine(cfg, dtype, device, max_batch=max_batch)

    def gpt_packed():
        p = pack.pack_gpt(synth.gpt_state_dict(cfg, 1234), cfg)
        return pack.quantize_gpt_fp8(p) if gpt_fp8 else p

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dp.replicate_packed(eng, [gpt_packed, lambda: pack.pack_bigvgan(synth.bigvgan_state_dict(cfg, 1234), cfg)])
    torch.cuda.synchronize()
    eng.replicate_s = time.perf_counter() - t0  # rank 0: synthesis + packing + broadcast; other ranks: wait + broadcast
    eng.finalize()
    return eng


def cpu_baseline(cfg, a):
    """Oracle (CPU port of the reference path, torch fp32, all host cores) on a bounded sample of the same
    workload; extrapolated per phase to one L/T sentence.  Reported next to the GPU number, never the target."""
    from itts_hip import synth
    from oracle import gpt as ogpt
    from oracle import vocoder as ovoc
    from itts_hip.config import ecapa_dims

    # "all cores" = the box's CPU SHARE, not the host's core count: an 8-GPU host shows 256 CPUs to os.cpu_count() and to
    # sched_getaffinity while a one-GPU box may use 16 of them (256 torch threads on 16 CPUs took 41 s for a 0.2 s phase)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    limited = False
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores, limited = min(cores, max(1, int(float(txt[0]) / float(txt[1]) + 0.5))), True
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores, limited = min(cores, max(1, int(q / per + 0.5))), True
            break
        except Exception:
            continue
    if not limited and cores > 32:
        cores = 16  # no quota visible: the documented share of a one-GPU box
    cores = max(1, int(os.environ.get("ITTS_CPU_BASELINE_THREADS", str(cores))))
    torch.set_num_threads(cores)
    print(f"[cpu_baseline] oracle on {cores} threads ...", file=sys.stderr, flush=True)
    g = cfg["gpt"]
    wg = ogpt.to_torch(synth.gpt_state_dict(cfg, 1234))
    wb = ogpt.to_torch(synth.bigvgan_state_dict(cfg, 1234))
    mel = torch.from_numpy(synth.prompt_mel(a.prompt_frames, seed=7))
    L, T = a.text_tokens, a.mel_tokens
    text = torch.from_numpy(synth.text_ids(L, 11, g["number_text_tokens"])).view(1, L)
    with torch.no_grad():
        t0 = time.perf_counter()
        cond = ogpt.get_conditioning(mel, wg, g)
        t_cond = time.perf_counter() - t0
        print(f"[cpu_baseline] conditioning {t_cond:.2f}s", file=sys.stderr, flush=True)
        nsamp = 24
        t0 = time.perf_counter()
        codes = ogpt.greedy_generate(cond, text, wg, g, nsamp, suppress_eos=True)
        t_gen = time.perf_counter() - t0
        print(f"[cpu_baseline] {nsamp} greedy steps {t_gen:.2f}s", file=sys.stderr, flush=True)
        # prefill alone, to separate it from the per-token cost
        t0 = time.perf_counter()
        ogpt.greedy_generate(cond, text, wg, g, 1, suppress_eos=True)
        t_prefill = time.perf_counter() - t0
        t_tok = (t_gen - t_prefill) / (nsamp - 1)
        full_codes = torch.from_numpy(synth.text_ids(T, 5, g["stop_mel_token"] - 1)).view(1, T)
        t0 = time.perf_counter()
        lat = ogpt.latent_forward(cond, text, full_codes, wg, g)
        t_lat = time.perf_counter() - t0
        print(f"[cpu_baseline] latent pass {t_lat:.2f}s", file=sys.stderr, flush=True)
        nfr = 16
        t0 = time.perf_counter()
        ovoc.bigvgan_forward(lat[:, :nfr], mel.transpose(1, 2), wb, cfg["bigvgan"], ecapa_dims(cfg["bigvgan"]))
        t_voc = time.perf_counter() - t0
    # reference recomputes conditioning twice per sentence (model.py:540,670)
    per_sentence = 2 * t_cond + t_prefill + (T - 1) * t_tok + t_lat + t_voc * (T / nfr)
    audio = T * 1024 / 24000.0
    return {
        "value": round(audio / per_sentence, 4), "unit": "audio_sec_per_sec", "cores": cores, "kind": "port",
        "sample": (f"oracle (torch fp32, {cores} threads): conditioning {t_cond:.2f}s x2, prefill s={32 + L + 2} {t_prefill:.2f}s, "
                   f"{nsamp} greedy steps -> {t_tok * 1e3:.1f} ms/token extrapolated to T={T}, latent pass T={T} {t_lat:.2f}s, "
                   f"BigVGAN+ECAPA on {nfr} frames {t_voc:.2f}s scaled x{T / nfr:.0f}"),
    }


def measure(eng, cfg, a, BU, steps, warmup, rank, world, product_loop=False):
    """Time `steps` passes of the hot path over this rank's shard (BU utterances per GPU, weak scaling) and return the
    aggregated numbers.  The global utterance list (world * BU equal-length utterances) is dealt by dp.partition; the
    data path has no collective."""
    from itts_hip import dp, synth
    from itts_hip.infer_core import remove_long_silence

    g = cfg["gpt"]
    device = eng.device
    L, T, NS = a.text_tokens, a.mel_tokens, a.sentences
    n_utt = world * BU
    mine = dp.partition([NS * L] * n_utt, world, rank)
    assert len(mine) == BU, (mine, BU)
    B = BU * NS
    mel = torch.from_numpy(synth.prompt_mel(a.prompt_frames, seed=7)).to(device)  # resident in HBM
    texts = np.stack([synth.text_ids(L, 11 + u * NS + k, g["number_text_tokens"]) for u in mine for k in range(NS)]).astype(np.int32)
    beam_uniforms = np.random.default_rng(5).random((T, B, 2 * a.beams), dtype=np.float32) if a.beams > 1 else None
    sample_uniforms = np.random.default_rng(6).random((T, B), dtype=np.float32) if a.sample else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    dec_ms, dec_steps = [0.0], [0]
    # per-phase device time (conditioning+ECAPA / prefill+AR decode / latent pass / vocoder), as infer.py:218-220 prints
    pev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    phase_ms = {"conditioning": 0.0, "ar_decode": 0.0, "latent": 0.0, "vocoder": 0.0}
    last = {}

    def step(timed: bool):
        if timed:
            pev[0].record(eng.stream)
        spk = eng.ecapa(mel.transpose(1, 2), overlap=True)  # as IndexTTS.infer: beside the conditioning encoder and the prefill
        cond = eng.conditioning(mel)
        if timed:
            pev[1].record(eng.stream)
        if a.beams > 1:
            eng.set_beam_sample(a.beams, 30, 0.8, 1.0, beam_uniforms)
        elif a.sample:
            eng.set_sampling(True, 30, 0.8, 1.0, sample_uniforms)
        eng.prefill(cond, texts, T, 10.0, True)
        if timed:
            ev[0].record(eng.stream)  # HIP events on the stream the decode graphs are launched on
        if product_loop:  # Engine.generate's loop with eos enabled: status() (sync + D2H) in front of every 16 steps
            done = 1
            while done < T:
                eng.status()
                n = min(16, T - done)
                eng.decode(n)
                done += n
        else:
            eng.decode(T - 1)
        if timed:
            ev[1].record(eng.stream)
        codes = eng.fetch()
        eng._exit()
        if a.beams > 1:
            eng.set_beam_sample(1)
        elif a.sample:
            eng.set_sampling(False)
        if timed:
            dec_ms[0] += ev[0].elapsed_time(ev[1])
            dec_steps[0] += T - 1
            pev[2].record(eng.stream)
        clean = []
        for i in range(B):
            c, n = remove_long_silence(codes[i:i + 1].astype(np.int64), g["stop_mel_token"])
            clean.append(c[0, :int(n[0])])
        lats = eng.latent_batch(cond, [texts[i] for i in range(B)], clean)
        if timed:
            pev[3].record(eng.stream)
        outs = eng.bigvgan_grouped(lats, spk)  # equal-length sentences share one batched launch sequence
        host = [torch.clamp(32767 * w, -32767.0, 32767.0).to(torch.int16).cpu() for w in outs]
        nsamp = sum(w.numel() for w in host)
        if timed:
            pev[4].record(eng.stream)
            pev[4].synchronize()
            for name, i in (("conditioning", 0), ("ar_decode", 1), ("latent", 2), ("vocoder", 3)):
                phase_ms[name] += pev[i].elapsed_time(pev[i + 1])
        last["wav"] = host
        return nsamp

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step(False)
    sync_all()
    t0 = time.perf_counter()
    samples = 0
    for _ in range(steps):
        samples += step(True)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        on_gpu = dist.get_backend() == "nccl"
        tot = torch.tensor([dt, float(samples)], dtype=torch.float64, device=device if on_gpu else "cpu")
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        samples = float(tot[1])
        # outside the timed region: the shard results travel to rank 0 once (dp.gather_waveforms), as a user of
        # dp.run_sharded gets them; checks that every utterance of the global list was synthesised exactly once
        per_utt = {u: np.concatenate([last["wav"][j * NS + k].numpy().reshape(-1) for k in range(NS)]) for j, u in enumerate(mine)}
        full = dp.gather_waveforms(per_utt, n_utt)
        if rank == 0:
            assert len(full) == n_utt and all(w.shape[0] == NS * T * 1024 for w in full)
    audio_s = samples / 24000.0
    # ---- roofline of the dominant kernel group: the per-token decode step ----
    D, NL, V = g["model_dim"], g["layers"], g["number_mel_codes"]
    esz = 2 if a.dtype in ("bf16", "f16") else 4
    w_params = NL * (12 * D * D + 13 * D) + 4 * D + D * V + V
    s_bar = (32 + L + 2 + 1) + T / 2.0
    kv_per_pos = 2 * NL * D * esz
    w_bytes = w_params * esz
    if a.gpt_fp8:  # the decode projections stream the fp8 copy (+ one fp32 scale per output row); biases stay fp32-sized
        w_bytes = NL * 12 * D * D + D * V + 4 * (NL * 9 * D + V) + esz * (NL * 13 * D + 4 * D + V)
    step_bytes = w_bytes + B * a.beams * kv_per_pos * s_bar
    ms_step = dec_ms[0] / max(dec_steps[0], 1)
    achieved = step_bytes / (ms_step * 1e-3) / 1e9
    # HBM traffic of the decode step: the committed rocprofv3 --pmc passes of this command (tools/round_profile.sh runs
    # `bench.py --no-graph` with FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 correction, at the mean sequence
    # length of this workload) - a separate run, as counters cannot be read inside a timed run; only quoted when rows,
    # mean S and dtype match it
    traffic, traffic_src = None, None
    try:
        pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_decode.json") or f.endswith("_pmc_decode_b32.json"))
        for f in reversed(pm):
            pj = json.load(open(os.path.join(ROOT, "profiles", f)))
            if (int(pj.get("decode_rows", -1)) == B * a.beams and abs(float(pj.get("mean_S", -1)) - s_bar) <= 2.0 and a.dtype == "bf16"
                    and not a.micro and not a.gpt_fp8):
                if int(pj.get("engine", 0)) != eng.decode_mode():
                    continue  # counters of the other decode path (persistent engine vs launches)
                traffic = int(pj["hbm_bytes_per_step"])
                traffic_src = (f"STATIC, from profiles/{f} (kernels at {pj.get('kernels_head', '?')}): rocprofv3 --pmc FETCH_SIZE / "
                               f"WRITE_SIZE passes of bench.py --no-graph at this run's mean sequence length (S = {s_bar:.0f}), "
                               f"2 x FETCH_SIZE + WRITE_SIZE per decode step; not re-measured by this run")
                break
    except Exception:
        traffic = None
    return {
        "value": round(audio_s / dt, 3), "ms_per_step": round(dt / steps * 1e3, 2), "rtf": round(dt / audio_s, 5),
        "phases_ms_per_step": {k: round(v / steps, 2) for k, v in phase_ms.items()},
        "config": {"workload": ("IndexTTS-1.5, %d utterance(s)/GPU x %d sentences x (L=%d text tokens, T=%d mel codes), "
                                "prompt %d frames, %s fixed-length decode, rep_penalty 10" %
                                (BU, NS, L, T, a.prompt_frames, ("sample" if a.sample else "greedy") if a.beams == 1 else "beam-sample x%d" % a.beams)
                                + (", product loop (status() every 16 steps)" if product_loop else "")),
                   "utterances_per_gpu": BU, "decode_batch": B * a.beams, "audio_sec_per_step_per_gpu": round(audio_s / steps / world, 3)},
        "roofline": {"bound": "hbm", "kernel": (("gpt decode step = decode_engine_kernel: ONE persistent launch for the 24 blocks, the head and the greedy sampler" if a.beams == 1 and not a.sample else
                                                 "gpt decode step (hipGraph: decode_engine_kernel - ONE persistent launch for the 24 blocks + head - then the sampler kernels)")
                                                if eng.decode_mode() == 1 else
                                                "gpt decode step (hipGraph: 97 gemv + 24 cache-attention + sampler)" if B <= 4 else
                                                "gpt decode step (hipGraph: 97 skinny MFMA gemm + 49 layernorm + 24 cache-attention + sampler)"),
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": int(step_bytes), "avg_launch_ms": round(ms_step, 4),
                     "decode_tokens_per_s": round(B * 1e3 / ms_step, 1)},
    }


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        raise SystemExit(launch_ranks(a))  # before anything in this process touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != max(a.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {a.gpus} does not match WORLD_SIZE={world}")
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ITTS_DIST_BACKEND", "nccl")  # "gloo": rehearse N ranks on a one-GPU box
        if backend != "nccl":
            local = local % max(torch.cuda.device_count(), 1)
            # ranks SHARE a GPU in this rehearsal mode: the persistent decode engine needs all 256 CUs of a device for its
            # one process (two co-running persistent grids could starve each other until their bounded waits give up)
            os.environ["ITTS_ENGINE"] = "0"
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    from itts_hip import config as icfg

    cfg = icfg.micro() if a.micro else icfg.indextts_1_5()
    if a.micro:
        a.text_tokens, a.mel_tokens, a.prompt_frames = 11, 24, 61
    device = f"cuda:{local}"
    eng = build_engine_dp(cfg, a.dtype, device, a.gpt_fp8)
    if a.no_graph:
        eng.debug(no_graph=True)
    m = measure(eng, cfg, a, a.batch, a.steps, a.warmup, rank, world, product_loop=a.eos)
    out = {
        "metric": "audio_sec_per_sec", "value": m["value"], "unit": "audio-s/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": ("fp8-e4m3 gpt weights (decode), bf16 activations/KV" if a.gpt_fp8 else a.dtype), "data": "synthetic",
        "rtf": m["rtf"], "phases_ms_per_step": m["phases_ms_per_step"], "config": m["config"], "roofline": m["roofline"],
    }
    out["decode_mode"] = "persistent_engine" if eng.decode_mode() == 1 else "launch_path"
    if world > 1:
        import torch.distributed as dist

        # every rank must have decoded on the path the headline describes: under nccl (one rank per GPU) that is the persistent
        # engine - a silent per-rank downgrade (hand-off timeout, fewer CUs on a partitioned device) must show in SCALE_rNN.json
        mode = torch.tensor([eng.decode_mode()], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        lo = mode.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                       "weights_replicate_s": round(getattr(eng, "replicate_s", 0.0), 3),
                       "collectives_in_timed_region": 0,
                       "ranks_on_persistent_engine": "all" if int(lo[0]) == 1 else "NOT all"}
        if dist.get_backend() == "nccl" and a.dtype in ("bf16", "f16") and not a.micro and os.environ.get("ITTS_ENGINE", "1") != "0":
            assert int(lo[0]) == 1, "a rank fell back from the persistent decode engine to the launch path"
    # the default run also measures, next to the headline line (the same per-GPU workload at every N, so the driver's
    # scaling curve compares like with like): 32 utterances per GPU = 64 decode rows - BASELINE config 3 at one GPU, BASELINE
    # config 4 (N x 32 utterances sharded data-parallel) at N GPUs - and, at one GPU, the product loop of Engine.generate
    # (status() read-back every 16 steps).  Never allowed to break the headline.
    if a.batch == 1 and not (a.no_also or a.no_graph or a.gpt_fp8 or a.eos) and a.dtype == "bf16" and a.beams == 1 and not a.sample:
        also = {}
        key = "config3_batch32" if world == 1 else "config4_batch32_per_gpu"
        try:
            m3 = measure(eng, cfg, a, 32, 2, 1, rank, world)
            also[key] = {"value": m3["value"], "unit": "audio-s/s", "n_gpus": world, "steps": 2, "warmup": 1,
                         "ms_per_step": m3["ms_per_step"], "phases_ms_per_step": m3["phases_ms_per_step"],
                         "config": m3["config"], "roofline": m3["roofline"]}
        except Exception as e:  # noqa: BLE001
            also[key] = {"error": repr(e)[:200]}
        if world == 1:
            try:
                mp = measure(eng, cfg, a, a.batch, max(2, min(a.steps, 5)), 1, rank, world, product_loop=True)
                also["product_loop"] = {"value": mp["value"], "unit": "audio-s/s", "ms_per_step": mp["ms_per_step"],
                                        "vs_headline": round(mp["value"] / max(m["value"], 1e-9), 4), "config": mp["config"],
                                        "decode_ms_per_token_step": mp["roofline"]["avg_launch_ms"]}
            except Exception as e:  # noqa: BLE001
                also["product_loop"] = {"error": repr(e)[:200]}
            try:  # the same utterance in the reference's DEFAULT generate() mode (infer.py:116-124: beam-sample, 3 beams): 6 decode rows
                import copy

                ab = copy.copy(a)
                ab.beams = 3
                mb = measure(eng, cfg, ab, a.batch, 2, 1, rank, world)
                also["reference_default_mode_3_beams"] = {"value": mb["value"], "unit": "audio-s/s", "ms_per_step": mb["ms_per_step"],
                                                          "vs_headline": round(mb["value"] / max(m["value"], 1e-9), 4), "config": mb["config"],
                                                          "decode_ms_per_token_step": mb["roofline"]["avg_launch_ms"],
                                                          "kernel": mb["roofline"]["kernel"]}
            except Exception as e:  # noqa: BLE001
                also["reference_default_mode_3_beams"] = {"error": repr(e)[:200]}
            try:  # BASELINE config 5: one 2000-char utterance = 20 sentences as one decode batch, fp8-e4m3 GPT weights (bf16 activations / KV)
                import copy

                a5 = copy.copy(a)
                a5.sentences, a5.gpt_fp8 = 20, True
                eng5 = build_engine_dp(cfg, a.dtype, device, True)
                m5 = measure(eng5, cfg, a5, 1, 2, 1, rank, world)
                also["config5_longform_fp8"] = {"value": m5["value"], "unit": "audio-s/s", "steps": 2, "warmup": 1, "ms_per_step": m5["ms_per_step"],
                                                "dtype": "fp8-e4m3 gpt weights (decode), bf16 activations/KV",
                                                "phases_ms_per_step": m5["phases_ms_per_step"], "config": m5["config"], "roofline": m5["roofline"]}
                del eng5
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                also["config5_longform_fp8"] = {"error": repr(e)[:200]}
            try:  # the reference's own GPU precision (is_fp16=True = IEEE half: libitts_hip_f16.so), same utterance, same kernels' f16 build
                eng16 = build_engine_dp(cfg, "f16", device)
                a16 = copy.copy(a)
                a16.dtype = "f16"
                m16 = measure(eng16, cfg, a16, a.batch, 3, 1, rank, world)
                also["ieee_half_f16"] = {"value": m16["value"], "unit": "audio-s/s", "steps": 3, "warmup": 1, "ms_per_step": m16["ms_per_step"],
                                         "vs_headline": round(m16["value"] / max(m["value"], 1e-9), 4), "dtype": "f16",
                                         "decode_ms_per_token_step": m16["roofline"]["avg_launch_ms"], "decode_mode": eng16.decode_mode()}
                del eng16
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                also["ieee_half_f16"] = {"error": repr(e)[:200]}
        out["also"] = also
    if rank == 0:
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
