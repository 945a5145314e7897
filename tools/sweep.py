"""Throughput sweep of the walk kernel over workloads / occupancies (run on the GPU box)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spumoni_amd import capi, synth

def run(tag, raw, seqs, offs, mode=capi.SPX_MODE_PML, docs=False, waves=0, reps=3, lpw=0, no_class=False):
    waves = waves or int(os.environ.get("SWEEP_WAVES", "0"))
    ix = capi.Index.from_raw(raw, 0)
    if waves: ix.set_option("waves_per_cu", waves)
    if lpw: ix.set_option("lanes_per_wave", lpw); tag = f"{tag} lanes/wave={lpw}"
    if os.environ.get("SWEEP_CHUNK_MODE"): ix.set_option("chunk_mode", int(os.environ["SWEEP_CHUNK_MODE"]))
    if os.environ.get("SWEEP_CHUNK_SHIFT"): ix.set_option("chunk_shift", int(os.environ["SWEEP_CHUNK_SHIFT"]))
    total = int(seqs.numel()); nreads = offs.numel() - 1
    d_seqs = capi.pad_seqs(seqs)
    d_len = torch.empty(total, dtype=torch.int32, device="cuda") if mode == capi.SPX_MODE_PML else None
    d_ptr = torch.empty(total, dtype=torch.int64, device="cuda") if mode == capi.SPX_MODE_MS else None
    d_doc = torch.empty(total, dtype=torch.int32, device="cuda") if docs else None
    d_cls = torch.empty((nreads, 2), dtype=torch.int64, device="cuda") if (mode == capi.SPX_MODE_PML and not no_class) else None
    ms = []
    for _ in range(reps + 1):
        ix.query_device(mode, d_seqs, offs, total, d_lengths=d_len, d_pointers=d_ptr, d_docs=d_doc, d_class=d_cls,
                        bin_width=150, max_value_thr=5)
        st = ix.last_stats(); ms.append(st["kernel_ms"])
    k = float(np.median(ms[1:]))
    f_mis = st["jumps"] / st["steps"]; f_pred = st["pred_jumps"] / st["steps"]
    out_b = 4 if mode == capi.SPX_MODE_PML else 8
    bstep = 64 * (1 + 2 * f_mis + f_pred) + 1 + out_b + (4 if docs else 0)
    print(f"{tag:42s} r={raw.r:>11d} n/r={raw.n/raw.r:6.1f} waves={waves or 'max':>3} kernel {k:8.2f} ms  "
          f"{st['steps']/k/1e6:7.2f} Gsteps/s  {nreads/k/1e3:8.1f} Mreads/s  f_mis {f_mis:.3f} rows/step {st['row_loads']/st['steps']:.2f} "
          f"dir/step {st['dir_loads']/st['steps']:.2f}  roofline {bstep*st['steps']/k/1e6/8000:.3f}  idx {ix.device_bytes/2**30:.1f} GiB "
          f"chunk {ix.last_chunk_stats()}", flush=True)
    ix.close()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "occ"):
    raw = synth.statistical_rlbwt(1 << 28, 253, 8.0, seed=3, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13)
    for w in (4, 8, 12, 16, 20):
        run("C3 minimizer sigma=253 m=44", raw, seqs, offs, waves=w)
    del raw, seqs, offs
if which in ("all", "dna"):
    raw = synth.statistical_rlbwt(1 << 27, 4, 60.0, seed=4, device="cuda", letters=b"ACGT")
    seqs, offs = synth.simulate_reads(raw, 4_000_000, 200, seed=14)
    run("C3 DNA sigma=4 mean_run=60 m=200", raw, seqs, offs)
    del raw, seqs, offs
if which in ("all", "ms"):
    raw = synth.statistical_rlbwt(1 << 27, 253, 8.0, seed=5, device="cuda", zipf=1.0, with_samples=True, n_docs=10)
    seqs, offs = synth.simulate_reads(raw, 5_000_000, 55, seed=15, warmup=int(os.environ.get("SWEEP_WARMUP", "4")))
    run("C4 MS + doc sigma=253 m=55", raw, seqs, offs, mode=capi.SPX_MODE_MS, docs=True)
    run("C4-shaped PML + doc", raw, seqs, offs, docs=True)
    del raw, seqs, offs
if which in ("all", "long"):
    raw = synth.statistical_rlbwt(1 << 27, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    s50, o50 = synth.simulate_reads(raw, 50_000, 2200, seed=16)
    s6, o6 = synth.simulate_reads(raw, 6_250, 2200, seed=17)
    for cm, sh in (("1", ""), ("0", ""), ("2", "6"), ("2", "7"), ("2", "8")):
        os.environ["SWEEP_CHUNK_MODE"] = cm
        os.environ["SWEEP_CHUNK_SHIFT"] = sh or "0"
        tag = {"1": "plain", "0": "auto"}.get(cm, f"chunk 2^{sh}")
        run(f"C5 50k x 2200 [{tag}]", raw, s50, o50)
        run(f"C5 share 6250 x 2200 [{tag}]", raw, s6, o6)
    os.environ.pop("SWEEP_CHUNK_MODE"); os.environ.pop("SWEEP_CHUNK_SHIFT")
    del raw, s50, o50, s6, o6
if which in ("long1",):  # one configuration, for rocprofv3 --kernel-trace --stats
    raw = synth.statistical_rlbwt(1 << 27, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    n1 = int(os.environ.get("LONG1_READS", "50000"))
    s50, o50 = synth.simulate_reads(raw, n1, 2200, seed=16)
    run(f"C5 {n1} x 2200 [auto]", raw, s50, o50, reps=5)
if which in ("long2",):  # is pass 1 slower per step than the plain walk of as many short reads?
    raw = synth.statistical_rlbwt(1 << 27, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    s1, o1 = synth.simulate_reads(raw, 860_000, 128, seed=16)
    os.environ["SWEEP_CHUNK_MODE"] = "1"
    run("plain 860k x 128", raw, s1, o1)
    s2, o2 = synth.simulate_reads(raw, 2_500_000, 44, seed=16)
    run("plain 2.5M x 44", raw, s2, o2)
    s50, o50 = synth.simulate_reads(raw, 50_000, 2200, seed=16)
    for w in (12, 16, 20, 24):
        os.environ["SWEEP_CHUNK_MODE"] = "0"
        run(f"chunked 50k x 2200 waves={w}", raw, s50, o50, waves=w)
    os.environ.pop("SWEEP_CHUNK_MODE")
if which in ("longdna",):
    # un-digested long reads: DNA alphabet, 10 kbp
    raw = synth.statistical_rlbwt(1 << 27, 4, 60.0, seed=4, device="cuda", letters=b"ACGT")
    s6, o6 = synth.simulate_reads(raw, 6_250, 10_000, seed=17, f_mis=0.08)
    for cm, sh in (("1", ""), ("0", ""), ("2", "7"), ("2", "8"), ("2", "9")):
        os.environ["SWEEP_CHUNK_MODE"] = cm
        os.environ["SWEEP_CHUNK_SHIFT"] = sh or "0"
        run(f"C5 DNA 6250 x 10000 [{cm}/{sh}]", raw, s6, o6)
    os.environ.pop("SWEEP_CHUNK_MODE"); os.environ.pop("SWEEP_CHUNK_SHIFT")
if which in ("big",):
    t0 = time.time()
    raw = synth.statistical_rlbwt(1_000_000_000, 253, 8.0, seed=3, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13)
    torch.cuda.synchronize(); print("gen", time.time() - t0, flush=True)
    run("C3 r=1e9 minimizer sigma=253 m=44", raw, seqs, offs)

if which in ("lpw",):
    raw = synth.statistical_rlbwt(1 << 26, 253, 8.0, seed=3, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, 2_000_000, 44, seed=13)
    for lpw in (1, 4, 16, 64):
        for w in (12, 32):
            run("C3-shape r=2^26, 2M reads", raw, seqs, offs, waves=w, lpw=lpw, reps=2)
if which in ("mix",):
    raw = synth.statistical_rlbwt(1 << 28, 253, 8.0, seed=3, device="cuda", zipf=1.0)
    for pf in (1.0, 0.5, 0.0):
        seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13, positive_fraction=pf)
        run(f"C3 sigma=253 m=44 positive_fraction={pf}", raw, seqs, offs)

if which in ("noclass",):
    raw = synth.statistical_rlbwt(1 << 28, 253, 8.0, seed=3, device="cuda", zipf=1.0)
    seqs, offs = synth.simulate_reads(raw, 10_000_000, 44, seed=13)
    run("C3 with classifier", raw, seqs, offs)
    run("C3 without classifier", raw, seqs, offs, no_class=True)
if which in ("realdna",):
    # real BWT: 10 haplotypes (1 % SNPs + indels vs a 20 Mbp random base genome) + reverse complements
    t0 = time.time()
    base = synth.random_genome(20_000_000, seed=1)
    genomes = [base] + [synth.mutate(base, seed=s) for s in range(2, 11)]
    text, doc_lengths = synth.pangenome_text(genomes)
    print(f"text {text.size/1e6:.0f} Mbp in {time.time()-t0:.1f}s", flush=True)
    t0 = time.time()
    raw = synth.index_from_text(torch.from_numpy(text).cuda(), doc_lengths=doc_lengths, with_samples=False)
    torch.cuda.synchronize(); print(f"index (SA + LCP + thresholds on the GPU) {time.time()-t0:.1f}s", flush=True)
    nreads, m = 5_000_000, 200
    seqs, offs = synth.sample_reads(text, nreads, m, seed=12)
    run("real BWT, 10-hap 20 Mbp pangenome, 5M x 200 bp", raw, torch.from_numpy(seqs).cuda(), torch.from_numpy(offs).cuda())
if which in ("fatdiv",):
    import subprocess
    raw = synth.statistical_rlbwt(1 << 28, 253, 8.0, seed=3, device="cuda", zipf=1.0)
    seqs5, offs5 = synth.simulate_reads(raw, 10_000_000, 44, seed=13, positive_fraction=0.5)
    seqs0, offs0 = synth.simulate_reads(raw, 10_000_000, 44, seed=13, positive_fraction=0.0)
    for rep in range(2):
        for d in (3, 6, 12, 24):
            os.environ["SPX_FAT_DIV"] = str(d)
            run(f"fat div={d} mix 0.5", raw, seqs5, offs5, reps=3)
            run(f"fat div={d} random", raw, seqs0, offs0, reps=3)
if which in ("realmin",):
    # the C3 pipeline end to end on REAL data at small scale: 10-haplotype pangenome, every sequence
    # digested (-m, k=4, w=11) on the GPU, BWT index of the digested text, 200 bp DNA reads digested
    # and walked on the device (spx_digest_batch_device -> spx_query_batch_device, nothing leaves HBM)
    t0 = time.time()
    base = synth.random_genome(20_000_000, seed=1)
    genomes = [base] + [synth.mutate(base, seed=s) for s in range(2, 11)]
    dig = capi.digester(0)
    parts = []
    for g in genomes:
        for seq in (g, synth.revcomp(g)):
            d, _ = dig.digest_host(capi.SPX_DIGEST_PROMOTED, 4, 11, seq, np.array([0, seq.size], dtype=np.uint64))
            parts.append(d.copy())
    dtext = np.concatenate(parts)
    text, _ = synth.pangenome_text(genomes)
    print(f"DNA text {text.size/1e6:.0f} Mbp -> digested text {dtext.size/1e6:.1f} M minimizers in {time.time()-t0:.1f}s", flush=True)
    t0 = time.time()
    raw = synth.index_from_text(torch.from_numpy(dtext).cuda(), with_samples=False)
    torch.cuda.synchronize(); print(f"index of the digested text: n={raw.n} r={raw.r} n/r={raw.n/raw.r:.2f} in {time.time()-t0:.1f}s", flush=True)
    nreads, m = 10_000_000, 200
    seqs, offs = synth.sample_reads(text, nreads, m, seed=12)
    d_seqs, d_offs = torch.from_numpy(seqs).cuda(), torch.from_numpy(offs).cuda()
    ix = capi.Index.from_raw(raw, 0)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for rep in range(3):
        e[0].record()
        d_d, d_do = ix.digest_device(capi.SPX_DIGEST_PROMOTED, 4, 11, d_seqs, d_offs, nreads * m)
        e[1].record()
        total = int(d_do[-1].item())
        d_len = torch.empty(total + 8, dtype=torch.int32, device="cuda")
        d_cls = torch.empty((nreads, 2), dtype=torch.int64, device="cuda")
        t1 = torch.cuda.Event(enable_timing=True); t1.record()
        ix.query_device(capi.SPX_MODE_PML, d_d, d_do, total, d_lengths=d_len, d_class=d_cls, bin_width=50, max_value_thr=5)
        e[2].record(); torch.cuda.synchronize()
        st = ix.last_stats()
        dg, wk = e[0].elapsed_time(e[1]), t1.elapsed_time(e[2])
        print(f"real minimizer index, {nreads} x {m} bp DNA reads: digest {dg:.2f} ms + walk {wk:.2f} ms "
              f"({total/nreads:.1f} minimizers/read, f_mis {st['jumps']/st['steps']:.3f}, {st['steps']/st['kernel_ms']/1e6:.1f} G steps/s) "
              f"-> {nreads/(dg+wk)/1e3:.1f} M reads/s = {nreads*m/(dg+wk)/1e6:.1f} G bases/s", flush=True)
    run("real minimizer index, digested reads (walk only)", raw, d_d[:total].clone(), d_do)
    for w in (12, 16):
        run("real minimizer index, digested reads (walk only)", raw, d_d[:total].clone(), d_do, waves=w)
if which in ("longlpw",):
    raw = synth.statistical_rlbwt(1 << 27, 253, 8.0, seed=6, device="cuda", zipf=1.0)
    for n in (6_250, 50_000):
        seqs, offs = synth.simulate_reads(raw, n, 2200, seed=17)
        for lpw in (0, 1, 2, 4, 8):
            run(f"C5 {n} x 2200", raw, seqs, offs, lpw=lpw, reps=2)
