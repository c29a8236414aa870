"""Host model of the scatter pass of the partitioned group-by (polars_amd/csrc/partition2_device.hpp: part2_scatter_body), written
with the SAME state variables and formulas as the kernel (fill / limit word, ring of 128-B lines, line-granular flushes into
256-record chunks handed out of a private region, retry of appends past the limit, final partial flush).  It checks the
invariants the kernel's correctness rests on for many record widths, ring sizes and skews: every record lands exactly once, a
ring position is never overwritten before it was flushed, every flush is a whole aligned line (except the final tail), chunk
fills add up.  This is a model of the algorithm, not of the HIP code -- the kernel itself is covered by the GPU tests."""
import numpy as np
import pytest

CHUNK_RECS = 256
NO_CHUNK = 0xFFFFFFFF


class Model:
    def __init__(self, n_parts, rec_words, ring_lines, chunks_per_wg):
        self.NP, self.RW, self.ring_lines = n_parts, rec_words, ring_lines
        self.ring_dw = ring_lines * 32
        self.chunk_dw = CHUNK_RECS * rec_words
        self.ring = np.full((n_parts, self.ring_dw), -1, np.int64)
        self.ring_dirty = np.zeros((n_parts, self.ring_dw), bool)          # written and not yet flushed
        first_limit = min(self.ring_dw // rec_words, CHUNK_RECS)
        self.fill = np.zeros(n_parts, np.int64); self.limit = np.full(n_parts, first_limit, np.int64)
        self.fdw = np.zeros(n_parts, np.int64); self.chunk = np.full(n_parts, NO_CHUNK, np.int64)
        self.next_chunk = 0
        self.chunks_per_wg = chunks_per_wg
        self.recs = np.full((chunks_per_wg, self.chunk_dw), -1, np.int64)
        self.chunk_part = np.full(chunks_per_wg, NO_CHUNK, np.int64); self.chunk_fill = np.zeros(chunks_per_wg, np.int64)
        self.flushes = []                                                  # (dst dword offset in chunk, n dwords)

    def append(self, p, rec):
        """one row's atomicAdd on {limit : fill} + ring write; False = past the limit (the row stays pending)"""
        pos = self.fill[p]; self.fill[p] += 1
        if pos >= self.limit[p]:
            return False
        d0 = pos * self.RW
        for w in range(self.RW):
            i = (d0 + w) & (self.ring_dw - 1)
            assert not self.ring_dirty[p, i], "ring position overwritten before it was flushed"
            self.ring[p, i] = rec[w]; self.ring_dirty[p, i] = True
        return True

    def flush_phase(self, final):
        for p in range(self.NP):
            lim = self.limit[p]
            fill = min(self.fill[p], lim)
            f_dw, ch = self.fdw[p], self.chunk[p]
            avail = fill * self.RW
            target = self.chunk_dw if fill >= CHUNK_RECS else (avail & ~31)
            nl = (target - f_dw) >> 5
            assert 0 <= nl <= self.ring_lines
            if (nl or (final and avail > f_dw)) and ch == NO_CHUNK:
                assert self.next_chunk < self.chunks_per_wg, "chunk region too small"
                ch = self.next_chunk; self.next_chunk += 1
                self.chunk_part[ch] = p
            for done in range(nl):
                src = (((f_dw >> 5) + done) & (self.ring_lines - 1)) << 5
                dst = f_dw + done * 32
                assert dst % 32 == 0 and dst + 32 <= self.chunk_dw
                assert self.ring_dirty[p, src:src + 32].all(), "flushing a line that is not completely written"
                self.recs[ch, dst:dst + 32] = self.ring[p, src:src + 32]
                self.ring_dirty[p, src:src + 32] = False
                self.flushes.append((dst, 32))
            f_dw += nl * 32
            if final and ch != NO_CHUNK:
                for w in range(f_dw, avail):
                    i = w & (self.ring_dw - 1)
                    self.recs[ch, w] = self.ring[p, i]; self.ring_dirty[p, i] = False
                self.chunk_fill[ch] = fill
            elif fill >= CHUNK_RECS and f_dw == self.chunk_dw:
                self.chunk_fill[ch] = CHUNK_RECS
                ch, f_dw, fill = NO_CHUNK, 0, 0
            self.limit[p] = min((f_dw + self.ring_dw) // self.RW, CHUNK_RECS)
            self.fill[p] = fill; self.fdw[p] = f_dw; self.chunk[p] = ch


@pytest.mark.parametrize("rec_words,ring_lines,n_parts,skew", [(4, 2, 64, 0.0), (3, 2, 64, 0.0), (3, 4, 32, 0.0), (2, 2, 16, 0.0), (5, 2, 8, 0.0), (13, 2, 8, 0.0),
                                                              (4, 2, 64, 0.7), (3, 4, 32, 0.95), (7, 8, 4, 1.0), (1, 2, 128, 0.3)])
def test_every_record_lands_exactly_once(rec_words, ring_lines, n_parts, skew):
    rng = np.random.default_rng(rec_words * 100 + ring_lines * 10 + n_parts)
    rows_per_round, rounds = 512, 40
    n = rows_per_round * rounds
    parts = rng.integers(0, n_parts, n)
    parts[rng.random(n) < skew] = 3 % n_parts                                # a hot partition
    m = Model(n_parts, rec_words, ring_lines, chunks_per_wg=n // CHUNK_RECS + n_parts + 2)
    iters = 0
    for rd in range(rounds):
        pending = list(range(rd * rows_per_round, (rd + 1) * rows_per_round))
        while True:
            rng.shuffle(pending)                                              # LDS atomics arrive in any order
            pending = [i for i in pending if not m.append(int(parts[i]), [i * 16 + w for w in range(rec_words)])]
            m.flush_phase(False)
            iters += 1
            if not pending:
                break
    m.flush_phase(True)
    assert not m.ring_dirty.any()
    seen = np.zeros(n, bool)
    for ch in range(m.next_chunk):
        p, cnt = m.chunk_part[ch], m.chunk_fill[ch]
        assert p != NO_CHUNK and 0 < cnt <= CHUNK_RECS
        r = m.recs[ch, : cnt * rec_words].reshape(cnt, rec_words)
        rows = r[:, 0] // 16
        assert (r == rows[:, None] * 16 + np.arange(rec_words)[None, :]).all(), "torn record"
        assert (parts[rows] == p).all() and not seen[rows].any()
        seen[rows] = True
    assert seen.all()
    assert (m.chunk_part[m.next_chunk:] == NO_CHUNK).all()
    # rings sized for the arrival rate rarely retry; a hot partition does (that is what the hot-key path is for)
    if skew == 0.0 and rows_per_round / n_parts <= (ring_lines * 32 / rec_words - 32 / rec_words) * 0.55:
        assert iters <= rounds * 1.5, iters
