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


# ---------------------------------------------------------------------------------------------- xabs.hip (round 4)
def _xswz(key):
    return ((key & 3) << 2) | ((0x78 >> (2 * ((key >> 2) & 3))) & 3)


def test_xabs_workgroup_map_covers_every_slot_and_split_once():
    """xabs_attn_kernel: workgroup id -> (split, slot) for a runtime split count.  Every (split, slot) of the batch exactly once, a
    group of 4 consecutive slots of one split on ONE XCD (id % 8: their 32-byte partial sectors share 128-byte lines), XCDs balanced to
    within one group, and the launcher's grid is the smallest multiple of 32 ids that holds the groups."""
    for batch in (1, 3, 4, 8, 20, 48, 63, 64, 100, 128):
        for S in (2, 3, 4):
            n_grp = (batch + 3) // 4
            grid = (n_grp * S + 7) // 8 * 32
            seen, per_xcd = {}, [set() for _ in range(8)]
            for wg in range(grid):
                xr, xq = wg & 7, wg >> 3
                grp = (xq >> 2) * 8 + xr
                sp = grp // n_grp
                b = ((grp - sp * n_grp) << 2) + (xq & 3)
                if sp >= S or b >= batch:
                    continue
                assert (sp, b) not in seen
                seen[(sp, b)] = wg
                per_xcd[xr].add((sp, b >> 2))
            assert len(seen) == S * batch, (batch, S)
            for (sp, b), wg in seen.items():
                assert seen[(sp, b & ~3)] & 7 == wg & 7
            sizes = [len(x) for x in per_xcd]
            assert max(sizes) - min(sizes) <= 1, (batch, S, sizes)
            assert grid % 32 == 0 and grid - 32 < n_grp * S * 4 <= grid


def test_xabs_key_splits_tile_the_1500_positions():
    for S in (2, 3, 4):
        NT = (1500 + 15) // 16
        edges = [sp * NT // S for sp in range(S + 1)]
        assert edges[0] == 0 and edges[-1] == NT and all(b - a >= 4 for a, b in zip(edges, edges[1:]))      # the prologue requests tiles 0 .. 3


def test_xabs_swizzle_is_conflict_free_for_both_lds_read_patterns():
    """LDS tile row = D * 2 bytes (a multiple of 512: every row starts at bank 0), 16-byte chunk c of row `key` stored at chunk
    c ^ xswz(key).  (1) S phase, ds_read_b128: lane = key | k group << 4 reads chunk (k-step * 4 + k group) of row key; the hardware
    serves the wave as 4 groups of 16 lanes (MI355X_MICROARCH.md), each must touch 16 distinct 16-byte bank
    groups (64 banks x 4 B = 16 chunks).  (2) P V phase, ds_read_b64_tr_b16: 32-lane halves, each lane 8 bytes: 32 distinct 8-byte bank
    pairs.  Both for every k-step / channel tile."""
    for D in (512, 768, 1024, 1280):
        rowb = D * 2
        assert rowb % 256 == 0
        # (1) lanes 0..15 of a group share the k group, differ in key
        for kstep in range(D // 32):
            for kg in range(4):
                chunks = set()
                for key in range(16):
                    c = kstep * 4 + kg
                    addr = (key & 7) * rowb + ((c ^ _xswz(key)) << 4)      # key >> 3 selects the half slot (4096-byte multiple: same bank)
                    chunks.add((addr >> 4) & 15)
                assert len(chunks) == 16, (D, kstep, kg)
        # (2) supplier lane: 16-lane group g16, index sl: key = (g16 >> 1) * 8 + (sl >> 2) (+ 4), chunk 4 c4 + (g16 & 1) * 2 + ((sl & 3) >> 1), byte (sl & 1) * 8
        for c4 in range(D // 32):
            for second in (0, 4):
                for half in (0, 1):                      # lanes 0..31, 32..63
                    pairs = set()
                    for lane in range(half * 32, half * 32 + 32):
                        g16, sl = lane >> 4, lane & 15
                        key = (g16 >> 1) * 8 + (sl >> 2) + second
                        c = c4 * 4 + (g16 & 1) * 2 + ((sl & 3) >> 1)
                        addr = (key & 7) * rowb + ((c ^ _xswz(key)) << 4) + (sl & 1) * 8
                        pairs.add((addr >> 3) & 31)
                    assert len(pairs) == 32, (D, c4, second, half)


def test_xabs_ring_of_seven_half_tiles_never_overwrites_a_resident_tile():
    """Half tile k = 2 tile + {0, 1} lives in ring slot k % 7.  At the end of loop iteration i (after barrier D) waves 4-7 request the
    second half of tile i + 3 and waves 0-3 the first half of tile i + 4; tiles i + 1 (P V next) and i + 2 (S next) and the in-flight
    halves of tile i + 3 must not share a slot with what is requested."""
    for n in (5, 23, 24, 31, 32, 47):
        live = {}                                               # ring slot -> half id
        def put(h):
            s = h % 7
            live[s] = h
        for h in (0, 1, 2, 3, 4, 5, 6):                         # prologue: tiles 0, 1, 2 and the first half of tile 3
            if h // 2 < n:
                put(h)
        for i in range(n):
            for t in (i, i + 1):                                # resident during iteration i: tile i (P V), tile i + 1 (S)
                if t < n:
                    assert live.get((2 * t) % 7) == 2 * t and live.get((2 * t + 1) % 7) == 2 * t + 1, (n, i, t)
            needed = {h for t in (i + 1, i + 2) if t < n for h in (2 * t, 2 * t + 1)}
            if i + 3 < n:
                needed.add(2 * (i + 3))
            new = [h for h in (2 * (i + 3) + 1, 2 * (i + 4)) if h // 2 < n]
            for h in new:
                victim = live.get(h % 7)
                assert victim is None or victim not in needed, (n, i, h, victim)
                assert victim is None or victim // 2 <= i, (n, i, h, victim)     # only tile i (or older) is overwritten
                put(h)


def test_gemm256_staged_epilogue_maps_cover_the_wave_tile_exactly_once(tmp_path):
    """csrc/epi_stage.h (round 5): the write / read maps that turn a wave's 128 x 64 accumulator tile through 16 KB of LDS, replayed for all
    64 lanes by tests/native/epi_stage_check.cpp (the SAME header the kernel includes): every element reaches its row-major place exactly
    once, every access stays inside the wave's slice and is aligned, every store instruction covers whole 128-byte lines."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "epi_stage_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "whisperkit_amd", "csrc"),
                    os.path.join(root, "tests", "native", "epi_stage_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "EPI_STAGE_OK" in out.stdout, out.stdout[-2000:]


def test_vector_layernorm_lanes_cover_every_row_once_with_whole_vectors():
    """csrc/layernorm.hip layernorm_v4_kernel (round 6): lane l owns channels 4 l .. 4 l + 3 of every 256-channel group (LN_MAXE / 4 = 5 groups): at every width the
    library runs (multiples of 4 up to 1280) the lanes' vectors tile the row exactly once, never straddle its end, and a wave instruction reads whole 16-byte / writes whole
    8-byte pieces that are aligned when the row base is (rows are d floats / d halves apart: d % 4 == 0 keeps both alignments)."""
    NV = 20 // 4
    for d in (128, 384, 512, 768, 1024, 1280):
        seen = np.zeros(d, np.int32)
        for i in range(NV):
            for lane in range(64):
                c = 4 * lane + 256 * i
                if c < d:
                    assert c + 3 < d and (c * 4) % 16 == 0 and (c * 2) % 8 == 0
                    seen[c:c + 4] += 1
        assert (seen == 1).all(), d
        assert (d * 4) % 16 == 0 and (d * 2) % 8 == 0
    assert 4 * 63 + 256 * (NV - 1) + 3 == 1279          # the last lane of the last group ends the widest row
