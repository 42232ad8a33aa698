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


def make_genomes_big(n_species, strains_per_species, len_lo, len_hi, divergence, seed, device, extra_genome=0):
    """The species/strain model of make_genomes for databases of 100+ Gbp (BASELINE configs[2]): the lengths
    are drawn first and the bases written in place into ONE buffer (torch.cat of the pieces would need the memory twice).
    extra_genome > 0 appends one random genome of that many bases as a species of its own (a chromosome of more than 2^17
    windows).  A different random stream than make_genomes.  Returns (bases uint8, seq_off int64 [n+1], species int64 [n])."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    acgt = _ACGT.to(device)
    L = torch.randint(len_lo, len_hi + 1, (n_species,), generator=g, device=device).cpu()
    lens = L.repeat_interleave(strains_per_species)
    if extra_genome > 0:
        lens = torch.cat([lens, torch.tensor([extra_genome])])
    off = torch.zeros(lens.numel() + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lens, 0)
    bases = torch.empty(int(off[-1]), dtype=torch.uint8, device=device)
    t = 0
    for sp in range(n_species):
        n = int(L[sp])
        anc = torch.randint(0, 4, (n,), generator=g, device=device, dtype=torch.uint8)
        for _ in range(strains_per_species):
            mut = torch.rand(n, generator=g, device=device) < divergence
            shift = torch.randint(1, 4, (n,), generator=g, device=device, dtype=torch.uint8)
            codes = torch.where(mut, (anc + shift) & 3, anc)
            o = int(off[t])
            torch.index_select(acgt, 0, codes.to(torch.int32), out=bases[o:o + n])
            t += 1
    species = torch.arange(n_species).repeat_interleave(strains_per_species)
    if extra_genome > 0:
        o = int(off[t])
        step = 1 << 26
        for a0 in range(0, extra_genome, step):
            m = min(step, extra_genome - a0)
            torch.index_select(acgt, 0, torch.randint(0, 4, (m,), generator=g, device=device, dtype=torch.int32), out=bases[o + a0:o + a0 + m])
        species = torch.cat([species, torch.tensor([n_species])])
    return bases, off.to(device), species.to(device)


def window_counts(seq_off, winlen=128, stride=113):
    """windows of every sequence (src/dna_encoding.h:259-276), int64 [n]"""
    n = seq_off[1:] - seq_off[:-1]
    nfull = torch.clamp(n - winlen, min=0) // stride + 1
    return torch.where(n <= winlen, torch.ones_like(n), nfull + (nfull * stride < n).to(n.dtype))


def add_genome(bases, seq_off, species, length, seed):
    """Appends one random genome of `length` bases as a species of its own (a chromosome with more than 2^17 windows
    -- 14.9 Mbp at the default stride -- makes a table RefSeq-like: target and window ids stop fitting 32 bits as fields)."""
    dev = bases.device
    g = torch.Generator(device=dev); g.manual_seed(seed)
    extra = _ACGT.to(dev)[torch.randint(0, 4, (length,), generator=g, device=dev, dtype=torch.int64)]
    off = torch.cat([seq_off, (seq_off[-1] + length).reshape(1)])
    sp = torch.cat([species, (species.max() + 1).reshape(1)])
    return torch.cat([bases, extra]), off, sp


def split_targets(seq_off, species, pieces, keep_last_whole=False):
    """Every genome becomes `pieces` targets of (nearly) equal length -- assemblies are many sequences per genome (RefSeq
    bacteria: ~2^15-2^17 sequences) -- of the same species.  The bases do not move.  Returns (seq_off, species)."""
    n = seq_off.numel() - 1
    dev = seq_off.device
    lens = seq_off[1:] - seq_off[:-1]
    j = torch.arange(pieces, device=dev, dtype=torch.int64)
    cut = seq_off[:-1, None] + (lens[:, None] * j[None, :]) // pieces          # [n, pieces] piece starts
    sp = species[:, None].expand(n, pieces)
    if keep_last_whole and n > 1:
        cut_l = torch.cat([cut[:-1].reshape(-1), seq_off[-2].reshape(1)])
        sp_l = torch.cat([sp[:-1].reshape(-1), species[-1].reshape(1)])
    else:
        cut_l, sp_l = cut.reshape(-1), sp.reshape(-1)
    return torch.cat([cut_l, seq_off[-1].reshape(1)]).contiguous(), sp_l.contiguous()


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


def sample_pairs(bases, seq_off, n_pairs, read_len, ins_lo, ins_hi, sub_rate, n_rate, seed):
    """Paired-end reads: fragment of length U[ins_lo, ins_hi]; mate 1 = its first read_len
    bases, mate 2 = the first read_len bases of its reverse complement.  Sequences 2q, 2q+1
    are the mates of pair q.  Returns (reads uint8, read_off int64 [2n+1], origin target)."""
    dev = bases.device
    g = torch.Generator(device=dev); g.manual_seed(seed)
    lens = (seq_off[1:] - seq_off[:-1]).to(torch.float64)
    w = torch.clamp(lens - ins_hi, min=0)
    tgt = torch.multinomial(w / w.sum(), n_pairs, replacement=True, generator=g)
    r = torch.rand(n_pairs, generator=g, device=dev, dtype=torch.float64)
    pos = seq_off[tgt] + (r * w[tgt]).to(torch.int64)
    ins = torch.randint(ins_lo, ins_hi + 1, (n_pairs,), generator=g, device=dev, dtype=torch.int64)
    ar = torch.arange(read_len, device=dev, dtype=torch.int64)[None, :]
    m1 = bases[pos[:, None] + ar]
    m2 = _comp_table(dev)[bases[(pos + ins - 1)[:, None] - ar].long()]
    reads = torch.stack([m1, m2], dim=1).reshape(2 * n_pairs, read_len)
    if sub_rate > 0:
        m = torch.rand(reads.shape, generator=g, device=dev) < sub_rate
        code = ((reads >> 1) & 3); code = code ^ (code >> 1)
        sh = torch.randint(1, 4, reads.shape, generator=g, device=dev, dtype=torch.uint8)
        reads = torch.where(m, _ACGT.to(dev)[((code + sh) & 3).long()], reads)
    if n_rate > 0:
        m = torch.rand(reads.shape, generator=g, device=dev) < n_rate
        reads = torch.where(m, torch.full_like(reads, 78), reads)
    off = torch.arange(2 * n_pairs + 1, device=dev, dtype=torch.int64) * read_len
    return reads.reshape(-1).contiguous(), off, tgt


def sample_long_reads(bases, seq_off, n_reads, mean_len, sub_rate, seed, min_len=500, max_len=60000, sigma=0.6):
    """ONT-like reads: lengths log-normal with the given mean (substitution errors only).
    Returns (reads uint8, read_off int64 [n+1], origin target)."""
    import math
    dev = bases.device
    g = torch.Generator(device=dev); g.manual_seed(seed)
    mu = math.log(mean_len) - 0.5 * sigma * sigma
    ln = torch.exp(mu + sigma * torch.randn(n_reads, generator=g, device=dev, dtype=torch.float64))
    ln = torch.clamp(ln, min_len, max_len).to(torch.int64)
    lens = (seq_off[1:] - seq_off[:-1])
    w = torch.clamp(lens - max_len, min=0).to(torch.float64)
    tgt = torch.multinomial(w / w.sum(), n_reads, replacement=True, generator=g)
    r = torch.rand(n_reads, generator=g, device=dev, dtype=torch.float64)
    pos = seq_off[tgt] + (r * (lens[tgt] - ln).to(torch.float64)).to(torch.int64)
    off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    torch.cumsum(ln, 0, out=off[1:])
    total = int(off[-1].item())
    rid = torch.repeat_interleave(torch.arange(n_reads, device=dev), ln)
    idx = pos[rid] + (torch.arange(total, device=dev, dtype=torch.int64) - off[rid])
    reads = bases[idx]
    del idx, rid
    if sub_rate > 0:
        m = torch.rand(total, generator=g, device=dev) < sub_rate
        code = ((reads >> 1) & 3); code = code ^ (code >> 1)
        sh = torch.randint(1, 4, (total,), generator=g, device=dev, dtype=torch.uint8)
        reads = torch.where(m, _ACGT.to(dev)[((code + sh) & 3).long()], reads)
    return reads.contiguous(), off, tgt
