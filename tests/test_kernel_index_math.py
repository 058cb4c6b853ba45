"""CPU emulation of index math the HIP kernels rely on (no GPU): claims made in kernel comments are checked here as arithmetic.

  * mel.hip (round 3): the signal under a workgroup's 64 frames sits in LDS skewed, sample j at j + 2 * (j // 160).  The MFMA loop forms
    x[n] +- x[400 - n] for frame rows (g * 16 + ai) from two ds_read_b32 at rowoff + o1 / rowoff + o2 - those addresses must be the
    skewed positions of samples row * 160 + n and row * 160 + 400 - n, inside the buffer, and conflict-free: a wave64 ds_read_b32 is
    served in two groups of 32 lanes (MI355X_MICROARCH.md, LDS table), bank = dword address mod 32.
"""
import numpy as np

K_HOP, K_NFFT, FG = 160, 400, 4
K_SPAN = (16 * FG - 1) * K_HOP + K_NFFT
K_SPAN_LDS = K_SPAN + 2 * (K_SPAN // K_HOP) + 2


def _pos(j):
    return j + 2 * (j // K_HOP)


def test_mel_span_skew_addresses_and_banks():
    assert K_SPAN == 10480
    assert max(_pos(j) for j in range(K_SPAN)) < K_SPAN_LDS
    assert len({_pos(j) for j in range(K_SPAN)}) == K_SPAN                     # injective: no two samples share a dword
    lanes = np.arange(64)
    ai, ak = lanes & 15, lanes >> 4
    for ks in range(50):
        k = ks * 4 + ak
        n = k + 1
        o1 = n + 2 * (n >= K_HOP)
        o2 = (K_NFFT - n) + 2 * (1 + ((K_NFFT - n) >= 2 * K_HOP))
        for g in range(FG):
            row = g * 16 + ai
            rowoff = row * (K_HOP + 2)
            a1, a2 = rowoff + o1, rowoff + o2
            assert np.array_equal(a1, np.array([_pos(r * K_HOP + nn) for r, nn in zip(row, n)]))
            assert np.array_equal(a2, np.array([_pos(r * K_HOP + K_NFFT - nn) for r, nn in zip(row, n)]))
            for addr in (a1, a2):
                for half in (slice(0, 32), slice(32, 64)):                       # the two 32-lane groups of a ds_read_b32
                    banks = addr[half] % 32
                    assert len(set(banks.tolist())) == 32, (ks, g, sorted(banks.tolist()))


def test_self_attention_row_bound_per_graph():
    """host.hip: one step graph per 8 positions, its self-attention launches fetch 1 + the largest position the graph reaches; the kernel
    instantiation covers ceil(rows / 32) passes of 32 rows and clamps the rest (decoder.hip dec_self_attn_kernel)."""
    k_max_tok, steps_per_graph = 224, 8
    seen = set()
    for first in range(0, 223, steps_per_graph):
        rows = min(max(first + steps_per_graph - 1, 0), k_max_tok - 1) + 1
        passes = (rows + 31) // 32
        assert 1 <= passes <= 7 and passes * 32 >= rows
        for pos in range(first, min(first + steps_per_graph, 223)):
            assert pos + 1 <= rows                                              # every position of the graph sees its whole history
        seen.add(rows)
    assert len(seen) == 28


def test_beam_row_owner_table_never_reads_a_row_another_history_overwrote():
    """beam.hip (round 3): no cache copies - row r of beam slot j's history is read from the cache of slot owner[j][r].  Simulation of the
    host bookkeeping (same rules as wh_decode_text_beam) with random re-parenting: pre-fill rows live in the audio's slot a (which is also
    beam slot a of audio a // beam: that beam only ever writes rows >= n_prompt - 1), every step writes row `t` of every live slot's own
    cache, a re-parented beam inherits the source's owner row.  Invariant: reading through the table returns exactly the sequence of
    (producer, row) pairs of the beam's true history."""
    rng = np.random.default_rng(5)
    for n_audio, beam, n_prompt, n_steps in ((3, 5, 4, 40), (2, 2, 1, 30), (1, 8, 6, 25)):
        n_slots, K = n_audio * beam, 224
        cache = {}                                              # (physical slot, row) -> what was written there
        for a in range(n_audio):                                # pre-fill on the audio slots: rows 0 .. n_prompt - 2
            for r in range(n_prompt - 1):
                cache[(a, r)] = ("prefill", a, r)
        owner = np.array([[(j // beam if r < n_prompt - 1 else j) for r in range(K)] for j in range(n_slots)])
        history = [[("prefill", j // beam, r) for r in range(n_prompt - 1)] for j in range(n_slots)]      # ground truth per beam slot
        for t in range(n_prompt - 1, n_prompt - 1 + n_steps):
            for j in range(n_slots):
                owner[j, t] = j
                assert (j, t) not in cache, "a slot writes each row once"
                cache[(j, t)] = ("step", j, t)
                history[j] = history[j] + [("step", j, t)]
            for j in range(n_slots):                            # what the self-attention kernel of this step reads
                got = [cache[(int(owner[j, r]), r)] for r in range(t + 1)]
                assert got == history[j], (n_audio, beam, t, j)
            new_owner, new_hist = owner.copy(), list(history)
            for a in range(n_audio):                            # ranking: every new beam continues some old beam of the same audio
                src = rng.integers(0, beam, size=beam)
                for j in range(beam):
                    s_, d_ = a * beam + int(src[j]), a * beam + j
                    new_owner[d_, : t + 1] = owner[s_, : t + 1]
                    new_hist[d_] = history[s_]
            owner, history = new_owner, new_hist
