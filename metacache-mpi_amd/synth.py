"""Seeded synthetic workloads for bench.py and the at-scale parity tests (SURVEY.md 8d:
there is no network, so RefSeq cannot be used; genomes follow a species/strain model so
that features are shared between strains and buckets hold several locations)."""
import torch

_ACGT = torch.tensor([65, 67, 71, 84], dtype=torch.uint8)          # A C G T
_COMP = None


def _comp_table(dev):
    global _COMP
    if _COMP is None or _COMP.device != dev:
        t = torch.arange(256, dtype=torch.uint8)
        for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
            t[a] = b
        _COMP = t.to(dev)
    return _COMP


def make_genomes(n_species, strains_per_species, len_lo, len_hi, divergence, seed, device):
    """Returns (bases uint8 [total], seq_off int64 [n+1], species_of_target int64 [n])."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    acgt = _ACGT.to(device)
    parts, lens, species = [], [], []
    for sp in range(n_species):
        L = int(torch.randint(len_lo, len_hi + 1, (1,), generator=g, device=device).item())
        anc = torch.randint(0, 4, (L,), generator=g, device=device, dtype=torch.int64)
        for _ in range(strains_per_species):
            mut = torch.rand(L, generator=g, device=device) < divergence
            shift = torch.randint(1, 4, (L,), generator=g, device=device, dtype=torch.int64)
            codes = torch.where(mut, (anc + shift) & 3, anc)
            parts.append(acgt[codes]); lens.append(L); species.append(sp)
    bases = torch.cat(parts)
    off = torch.zeros(len(lens) + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64, device=device), 0)
    return bases, off, torch.tensor(species, dtype=torch.int64, device=device)


def sample_reads(bases, seq_off, n_reads, read_len, sub_rate, n_rate, seed, revcomp_half=True):
    """Fixed-length reads drawn uniformly over all targets.  Returns (reads uint8 [n_reads*read_len],
    read_off int64 [n_reads+1], origin target int64 [n_reads])."""
    dev = bases.device
    g = torch.Generator(device=dev); g.manual_seed(seed)
    lens = (seq_off[1:] - seq_off[:-1]).to(torch.float64)
    w = torch.clamp(lens - read_len, min=0)
    tgt = torch.multinomial(w / w.sum(), n_reads, replacement=True, generator=g)
    r = torch.rand(n_reads, generator=g, device=dev, dtype=torch.float64)
    pos = seq_off[tgt] + (r * w[tgt]).to(torch.int64)
    idx = pos[:, None] + torch.arange(read_len, device=dev, dtype=torch.int64)[None, :]
    reads = bases[idx]                                                   # [n, L] uint8
    del idx
    if sub_rate > 0:
        m = torch.rand(reads.shape, generator=g, device=dev) < sub_rate
        # substitute by a different base: rotate within ACGT
        code = ((reads >> 1) & 3); code = code ^ (code >> 1)
        sh = torch.randint(1, 4, reads.shape, generator=g, device=dev, dtype=torch.uint8)
        newc = (code + sh) & 3
        reads = torch.where(m, _ACGT.to(dev)[newc.long()], reads)
    if n_rate > 0:
        m = torch.rand(reads.shape, generator=g, device=dev) < n_rate
        reads = torch.where(m, torch.full_like(reads, 78), reads)        # 'N'
    if revcomp_half:
        flip = torch.rand(n_reads, generator=g, device=dev) < 0.5
        rc = _comp_table(dev)[reads.flip(1).long()]
        reads = torch.where(flip[:, None], rc, reads)
    off = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * read_len
    return reads.reshape(-1).contiguous(), off, tgt
