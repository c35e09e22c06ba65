"""Synthetic PacBio HiFi reads with kinetics (stand-in for the unavailable tutorial / 30x BAMs).

Follows SURVEY.md section 8(d): i.i.d. genome with GC fraction g, reads with log-normal length
(median 15 kb, clipped to [1 kb, 30 kb]), random strand, unaligned (flag 4), per-base codev1
kinetics bytes `fi,fp,ri,rp` (IPD ~ round(Gamma(2,12)), PW ~ round(Gamma(3,5)), clipped to 0..255),
a small fraction of reads with `B:S` u16 frame arrays, a few short reads and reads with a missing tag
(the pass-through paths of reference src/app/hifimeth/mod_main.cpp:189-196).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

_NIB = np.array([1, 2, 4, 8, 15], np.uint8)  # A C G T N  (BAM seq_nt16 codes)
_ASCII = np.frombuffer(b"ACGTN", np.uint8)


@dataclass
class Read:
    """One unaligned read as the BAM record holds it."""
    name: str
    l_qseq: int
    flag: int
    seq4: np.ndarray                 # uint8[(L+1)//2], high nibble first
    fi: Optional[np.ndarray]         # uint8[L] (B:C) or uint16[L] (B:S); None = tag missing
    fp: Optional[np.ndarray]
    ri: Optional[np.ndarray]
    rp: Optional[np.ndarray]

    def has_kinetics(self) -> bool:
        return all(x is not None and len(x) == self.l_qseq for x in (self.fi, self.fp, self.ri, self.rp))

    def ascii(self) -> bytes:
        """Sequence as stored (not strand-normalised)."""
        return _ASCII[unpack_codes(self.seq4, self.l_qseq)].tobytes()


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """codes 0..4 (A,C,G,T,N) -> BAM 4-bit packed."""
    nib = _NIB[codes]
    if len(nib) & 1:
        nib = np.concatenate([nib, np.zeros(1, np.uint8)])
    return ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8)


def unpack_codes(seq4: np.ndarray, L: int) -> np.ndarray:
    nib = np.empty(2 * len(seq4), np.uint8)
    nib[0::2] = seq4 >> 4
    nib[1::2] = seq4 & 15
    lut = np.full(16, 4, np.uint8)
    lut[[1, 2, 4, 8]] = [0, 1, 2, 3]
    return lut[nib[:L]]


def read_from_ascii(seq: bytes, fi, fp, ri, rp, flag: int = 4, name: str = "r") -> Read:
    lut = np.full(256, 4, np.uint8)
    lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
    codes = lut[np.frombuffer(seq, np.uint8)]
    return Read(name, len(seq), flag, pack_codes(codes), fi, fp, ri, rp)


def _kinetics(rng, L: int, wide: bool):
    if not wide:
        return [np.clip(np.rint(rng.gamma(k, th, L)), 0, 255).astype(np.uint8)
                for k, th in ((2.0, 12.0), (3.0, 5.0), (2.0, 12.0), (3.0, 5.0))]
    # B:S arrays hold raw frame counts (may exceed 952; re-encoded lossily, bam_info.cpp:455-478)
    return [np.clip(np.rint(rng.gamma(2.0, s, L)), 0, 2000).astype(np.uint16) for s in (30.0, 12.0, 30.0, 12.0)]


def synth_reads(n_reads: int, seed: int = 20250220, gc: float = 0.36, median_len: int = 15000,
                sigma: float = 0.35, min_len: int = 1000, max_len: int = 30000, frac_wide: float = 0.01,
                frac_short: float = 0.005, frac_missing: float = 0.001, frac_n: float = 0.0,
                genome_len: int = 4_000_000) -> List[Read]:
    rng = np.random.default_rng(seed)
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    genome = rng.choice(4, size=genome_len, p=p).astype(np.uint8)
    out = []
    for i in range(n_reads):
        L = int(np.clip(rng.lognormal(np.log(median_len), sigma), min_len, max_len))
        if rng.random() < frac_short:
            L = int(rng.integers(50, min_len))
        L = min(L, genome_len)
        st = int(rng.integers(0, genome_len - L + 1))
        codes = genome[st:st + L].copy()
        if rng.random() < 0.5:  # read from the reverse strand, still stored as-is (unaligned, flag 4)
            codes = (3 - codes)[::-1].copy()
        if frac_n > 0:
            codes[rng.random(L) < frac_n] = 4
        fi, fp, ri, rp = _kinetics(rng, L, rng.random() < frac_wide)
        if rng.random() < frac_missing:
            rp = None
        out.append(Read(f"m0/{i}/ccs", L, 4, pack_codes(codes), fi, fp, ri, rp))
    return out


def expected_sites_per_base(gc: float) -> float:
    """CpG + CHG (fwd only) + CHH (both strands) density for i.i.d. bases."""
    c = gc / 2
    return c * c + c * c * (1 - c) + 2 * c * (1 - c) ** 2
