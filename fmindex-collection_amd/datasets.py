"""Texts to index when no real assembly is at hand, and the reference's FASTA rule for when one is.

* genome_like_text — a seeded repeat-structured stand-in for a mammalian assembly (bench.py's default text): uniform bases would make every
  suffix interval collapse to one row after ~log4(n) symbols, which is not what GRCh38 does.  Layers, applied in this order on a uniform
  {A,C,G,T} background:
    1. interspersed repeats: ~10^3 families (consensus 300 bp - 6 kbp, log-uniform; copy numbers Zipf-like), copies 5'-truncated to 25-100 %
       of the consensus, either strand, each family with its own divergence in [5 %, 20 %] (substitutions) — 45 % of the text;
    2. tandem satellites: arrays of 10^5 - 10^6 bp of a 5 - 2 000 bp unit at 2 % divergence — ~1 % of the text;
    3. runs of one symbol: per sequence a centromere-sized run, telomeric runs and a few gaps, written as rank 1 — the reference's loader
       turns N into A under --convertUnknownChar at sigma = 5 (src/example/utils.h:86-98) — 5 % of the text.
  Everything is drawn from one torch generator on the given device, so a (seed, lengths, device type) triple names one text.
* load_fasta — the reference example's reader (src/example/utils.h:26-104) for FMGPU_FASTA=<path>: '>' lines start a sequence, A/C/G/T
  (either case) -> 1..4, '$' -> 0, newlines dropped, every other byte -> 1 (convertUnknownChar, sigma = 5); the last byte of the file is
  never a symbol (the reader's end-of-file test comes first).
"""
import math

import numpy as np

__all__ = ["genome_like_text", "load_fasta", "GENOME_LIKE_DEFAULTS"]

GENOME_LIKE_DEFAULTS = dict(repeat_fraction=0.45, families_per_gbp=330.0, consensus_min=300, consensus_max=6000, divergence_min=0.05, divergence_max=0.20,
                            satellite_fraction=0.01, satellite_min=100_000, satellite_max=1_000_000, satellite_divergence=0.02,
                            run_fraction=0.05)


def genome_like_text(lengths, seed=42, device="cuda", **overrides):
    """returns (text uint8 tensor of sum(lengths) symbols in 1..4, stats dict)"""
    import torch
    P = dict(GENOME_LIKE_DEFAULTS); P.update(overrides)
    dev = torch.device(device)
    total = int(sum(lengths))
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    text = torch.empty(total, dtype=torch.uint8, device=dev)
    chunk = 1 << 28
    for lo in range(0, total, chunk):
        hi = min(total, lo + chunk)
        text[lo:hi] = torch.randint(1, 5, (hi - lo,), generator=g, device=dev, dtype=torch.uint8)
    stats = {"symbols": total}

    def mutate(block, prob):
        """substitute every position with its own probability by one of the three other bases"""
        hit = torch.rand(block.numel(), generator=g, device=dev) < prob
        shift = torch.randint(1, 4, (block.numel(),), generator=g, device=dev, dtype=torch.uint8)
        return torch.where(hit, (block - 1 + shift) % 4 + 1, block)

    # ---- 1. interspersed repeats
    nfam = max(4, int(round(P["families_per_gbp"] * total / 1e9)))
    lo_c, hi_c = P["consensus_min"], min(P["consensus_max"], max(P["consensus_min"] + 1, total // 50))
    cons_len = torch.exp(torch.rand(nfam, generator=g, device=dev) * (math.log(hi_c) - math.log(lo_c)) + math.log(lo_c)).long().clamp_(lo_c, hi_c)
    cons_off = torch.cumsum(cons_len, 0) - cons_len
    bank = torch.randint(1, 5, (int(cons_len.sum().item()),), generator=g, device=dev, dtype=torch.uint8)
    fam_div = torch.rand(nfam, generator=g, device=dev) * (P["divergence_max"] - P["divergence_min"]) + P["divergence_min"]
    weights = 1.0 / torch.arange(1, nfam + 1, device=dev, dtype=torch.float64) ** 0.9
    weights = weights[torch.randperm(nfam, generator=g, device=dev)]
    target = int(P["repeat_fraction"] * total)
    mean_len = float((cons_len.double() * weights).sum().item() / weights.sum().item()) * 0.625
    ncopies = int(target / max(mean_len, 1.0) * 1.3) + 16
    fam = torch.multinomial(weights, ncopies, replacement=True, generator=g)
    frac = torch.rand(ncopies, generator=g, device=dev) * 0.75 + 0.25
    clen = (cons_len[fam].double() * frac.double()).long().clamp_(min=20)
    keep = int((torch.cumsum(clen, 0) <= target).sum().item())
    fam, clen = fam[:keep], clen[:keep]
    start = (torch.rand(keep, generator=g, device=dev, dtype=torch.float64) * (total - clen).double()).long()
    strand = torch.rand(keep, generator=g, device=dev) < 0.5
    per = max(16, min(1 << 16, int(0.03 * total / max(mean_len, 1.0))))      # copies per pass: each pass covers ~3 % of the text
    written_rep = 0
    for c0 in range(0, keep, per):
        c1 = min(keep, c0 + per)
        # inside a pass the copies are put in text order and cut where the next one starts: no position is written twice, so the text does
        # not depend on the order in which the device performs the scattered stores (later passes overwrite earlier ones, like younger insertions)
        order = torch.argsort(start[c0:c1])
        st_p, fam_p, str_p, full = start[c0:c1][order], fam[c0:c1][order], strand[c0:c1][order], clen[c0:c1][order]
        nxt = torch.cat([st_p[1:], torch.full((1,), total, device=dev, dtype=st_p.dtype)])
        L = torch.minimum(full, nxt - st_p)
        first = torch.cumsum(L, 0) - L
        owner = torch.repeat_interleave(torch.arange(c1 - c0, device=dev), L)
        off = torch.arange(int(L.sum().item()), device=dev) - first[owner]
        f = fam_p[owner]
        end = cons_off[f] + cons_len[f]                      # a copy is the 3' end of its consensus (5' truncation)
        fwd = ~str_p[owner]
        src = torch.where(fwd, end - full[owner] + off, end - 1 - off)
        base = bank[src]
        base = torch.where(fwd, base, 5 - base)              # the other strand: reverse complement
        base = mutate(base, fam_div[f])
        text[st_p[owner] + off] = base
        written_rep += int(L.sum().item())
    stats["repeat_families"] = nfam
    stats["repeat_copies"] = keep
    stats["repeat_fraction_written"] = written_rep / total

    # ---- 2. tandem satellites
    sat_target = int(P["satellite_fraction"] * total)
    smax = max(1000, min(P["satellite_max"], total // 100))
    smin = min(P["satellite_min"], max(500, smax // 10))
    written, nsat = 0, 0
    while written < sat_target and nsat < 4096:
        ln = int(math.exp(float(torch.rand(1, generator=g, device=dev).item()) * (math.log(smax) - math.log(smin)) + math.log(smin)))
        ln = min(ln, sat_target - written + smin)
        unit_len = int(math.exp(float(torch.rand(1, generator=g, device=dev).item()) * (math.log(2000) - math.log(5)) + math.log(5)))
        unit = torch.randint(1, 5, (unit_len,), generator=g, device=dev, dtype=torch.uint8)
        pos = int(float(torch.rand(1, generator=g, device=dev, dtype=torch.float64).item()) * (total - ln))
        arr = unit[torch.arange(ln, device=dev) % unit_len]
        text[pos: pos + ln] = mutate(arr, torch.full((ln,), P["satellite_divergence"], device=dev))
        written += ln; nsat += 1
    stats["satellite_arrays"] = nsat
    stats["satellite_fraction_written"] = written / total

    # ---- 3. runs of one symbol (unknown bases read as A)
    run_total = 0
    base_off = 0
    for Ls in lengths:
        Ls = int(Ls)
        if Ls >= 2000:
            budget = int(P["run_fraction"] * Ls)
            telo = min(10_000, Ls // 200)
            cen = int(budget * 0.6)
            cpos = base_off + int(Ls * 0.4)
            text[cpos: cpos + cen] = 1
            text[base_off: base_off + telo] = 1
            text[base_off + Ls - telo: base_off + Ls] = 1
            rest = budget - cen - 2 * telo
            ngaps = max(1, Ls // 20_000_000 + 1)
            if rest > 0:
                gl = rest // ngaps
                gp = torch.rand(ngaps, generator=g, device=dev, dtype=torch.float64)
                for k in range(ngaps):
                    p = base_off + int(float(gp[k].item()) * (Ls - gl))
                    text[p: p + gl] = 1
            run_total += budget
        base_off += Ls
    stats["run_fraction_written"] = run_total / total
    return text, stats


_FASTA_LUT = None


def load_fasta(path, sigma=5):
    """the reference example's FASTA reader (src/example/utils.h:26-104) with --convertUnknownChar -> (symbols uint8 [total], seq_off int64 [nseq + 1])"""
    global _FASTA_LUT
    if sigma != 5:
        raise ValueError("the DNA reader is the sigma = 5 form")
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size == 0 or raw[0] != ord(">"):
        raise ValueError("can't read fasta file")          # utils.h:39-41
    nl_pos = np.nonzero(raw == ord("\n"))[0]
    # a '>' met while reading a sequence starts a name, which runs up to and including the next newline (utils.h:43-58, :64)
    hdr_start, hdr_end, until = [], [], 0
    for p in np.nonzero(raw == ord(">"))[0]:
        if p < until:
            continue                                         # a '>' inside a name is part of the name
        k = int(np.searchsorted(nl_pos, p))
        until = int(nl_pos[k]) + 1 if k < nl_pos.size else raw.size
        hdr_start.append(int(p)); hdr_end.append(until)
    hdr_start, hdr_end = np.asarray(hdr_start, dtype=np.int64), np.asarray(hdr_end, dtype=np.int64)
    in_hdr = np.zeros(raw.size + 1, dtype=np.int32)
    np.add.at(in_hdr, hdr_start, 1)
    np.add.at(in_hdr, hdr_end, -1)
    in_hdr = np.cumsum(in_hdr[:-1]) > 0
    body = raw
    nl = raw == ord("\n")
    last = np.zeros(raw.size, dtype=bool); last[-1] = True   # the byte before end-of-file closes the last record (utils.h:64-77) and is never a symbol
    has_seq = hdr_end < raw.size                             # a name that runs to the end of the file is followed by no sequence at all
    if _FASTA_LUT is None:
        lut = np.full(256, 1, dtype=np.uint8)               # convertUnknownChar at sigma = 5: rank 1 (utils.h:86-98)
        lut[ord("$")] = 0
        for ch, r in (("A", 1), ("C", 2), ("G", 3), ("T", 4)):
            lut[ord(ch)] = r; lut[ord(ch.lower())] = r
        _FASTA_LUT = lut
    keep = ~in_hdr & ~nl & ~last
    sym = _FASTA_LUT[body[keep]]
    # sequence k = the kept bytes between header k's end and header k+1's start
    kept_before = np.concatenate([[0], np.cumsum(keep)])
    seq_off = np.concatenate([kept_before[hdr_end[has_seq]], [kept_before[-1]]]).astype(np.int64)
    return sym, seq_off
