"""Restatement of the reference's MM/ML/MN tag builder (test infrastructure only).

Follows src/corelib/build_mod_bam.cpp:125-248: MM:Z = "C+m" {",d"} ";" "G-m" {",d"} ";" where d is the number of
unmodified-candidate bases (C on the forward strand, resp. G) skipped since the previous call; ML:B:C = forward calls
then reverse calls; MN = l_qseq.  `fwd_seq` is the forward-strand sequence (bam_info.cpp:224-233).

Pin status: the MM / ML strings these rules write are mapped back to exactly the calls they were written from by the
reference's own parser core (s_parse_one_mod_list, compiled in place into oracle/_ref/ref_modparse; asserted when
tests/golden/modparse.json is made) and equal hand-worked strings (modtags_known_answers.json); MN and the order / removal
of tags in the record follow build_mod_bam.cpp by reading only (its code calls the htslib library inline)."""
import numpy as np


def expected_tags(fwd_seq: bytes, qoff, strand, ml):
    qoff, strand, ml = np.asarray(qoff), np.asarray(strand), np.asarray(ml, np.uint8)
    if len(qoff) == 0:
        return None
    seq = np.frombuffer(fwd_seq, np.uint8)
    parts, mls = [], []
    for s, base, head in ((0, ord("C"), "C+m"), (1, ord("G"), "G-m")):
        sel = np.nonzero(strand == s)[0]
        sel = sel[np.argsort(qoff[sel], kind="stable")]
        q = qoff[sel]
        assert (np.diff(q) > 0).all() and (seq[q] == base).all()
        is_b = np.concatenate([[0], np.cumsum(seq == base)])          # prefix count of candidate bases
        last = np.concatenate([[0], q[:-1] + 1])
        deltas = is_b[q] - is_b[last]
        parts.append(head + "".join(f",{int(d)}" for d in deltas) + ";")
        mls.append(ml[sel])
    return dict(MM="".join(parts), ML=np.concatenate(mls), MN=len(fwd_seq))


def resolve_threshold(bins):
    """s_resolve_scaled_prob_threshold for one context (src/app/hifimeth/pileup.cpp:355-436):
    window [20, 236) trimmed on both sides while a bin holds < 10 samples; if the window is >= 50 bins wide the
    threshold is the FIRST minimum bin of the window, else / with < 10000 samples in the window it is 128.
    Returns (threshold, samples_in_window)."""
    a = [int(x) for x in bins]
    st, en = 20, 256 - 20
    while st < 256 and a[st] < 10:
        st += 1
    while en and a[en - 1] < 10:
        en -= 1
    total, min_i, min_cnt = 0, -1, None
    if en - st >= 50:
        for i in range(st, en):
            total += a[i]
            if min_cnt is None or min_cnt > a[i]:
                min_cnt, min_i = a[i], i
    if total < 10000 or min_i == -1:
        return 128, total
    return min_i, total
