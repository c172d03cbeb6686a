"""The synthetic baseline streams of the corner-feature tests: features the three bundled
x264 streams never use.  Every entry is a keyword set for tests/h264writer.StreamWriter; the reference's answers for
them are in tests/golden/synth_golden.json (tests/golden/make_synth_golden.py)."""
from h264writer import random_config

CONFIGS = {
    "plain_ip":            dict(n_pics=12, wmb=6, hmb=5, seed=2),
    "multi_slice_idc012":  dict(n_pics=10, wmb=5, hmb=4, seed=3, slices_per_pic=3, idc=(0, 1, 2)),
    "ipcm":                dict(n_pics=8, seed=4, p_pcm=0.2, slices_per_pic=2),
    "constrained_intra":   dict(n_pics=10, seed=5, constrained_intra=1, wmb=6, hmb=4, slices_per_pic=2),
    "multi_ref":           dict(n_pics=12, seed=6, num_ref_frames=3, num_ref_idx_active=3, wmb=5, hmb=4),
    "ref_list_reordering": dict(n_pics=12, seed=7, num_ref_frames=4, num_ref_idx_active=2, reorder=True, wmb=5, hmb=4),
    "fmo_interleaved":     dict(n_pics=10, seed=8, wmb=6, hmb=4, fmo=dict(type=0, groups=3, run_length=[3, 5, 2]), slices_per_pic=2),
    "fmo_dispersed":       dict(n_pics=8, seed=9, wmb=6, hmb=4, fmo=dict(type=1, groups=4)),
    "fmo_foreground":      dict(n_pics=8, seed=10, wmb=6, hmb=5, fmo=dict(type=2, groups=3, rects=[(1, 14), (16, 28)])),
    "fmo_raster_scan":     dict(n_pics=8, seed=11, wmb=6, hmb=5, fmo=dict(type=4, groups=2, direction=0, rate=7)),
    "fmo_wipe":            dict(n_pics=8, seed=12, wmb=6, hmb=5, fmo=dict(type=5, groups=2, direction=1, rate=4)),
    "fmo_explicit":        dict(n_pics=8, seed=13, wmb=4, hmb=3, fmo=dict(type=6, groups=3, ids=[0, 1, 2, 0, 2, 2, 1, 1, 0, 0, 1, 2])),
    "aso":                 dict(n_pics=10, seed=14, wmb=5, hmb=4, slices_per_pic=4, aso=True),
    "mmco_long_term":      dict(n_pics=30, seed=19, num_ref_frames=5, num_ref_idx_active=4, mmco=True, reorder=True, poc_type=0),
    "poc0_display_reorder": dict(n_pics=12, seed=16, poc_type=0, poc_pattern=[0, 3, 1, 2], num_ref_frames=3, wmb=5, hmb=4),
    "poc1_nonref_idr":     dict(n_pics=12, seed=17, poc_type=1, num_ref_frames=2, non_ref_every=3, idr_period=5),
    "frame_num_gaps":      dict(n_pics=20, seed=18, num_ref_frames=3, num_ref_idx_active=2, gaps=1, reorder=True),
    "gaps_full_dpb":       dict(wmb=5, hmb=6, n_pics=23, seed=29, poc_type=0, num_ref_frames=1, num_ref_idx_active=1, slices_per_pic=3,
                                idc=(2,), chroma_qp_offset=-7, aso=True, non_ref_every=4, gaps=1, reorder=True),
    "everything":          dict(n_pics=40, seed=20, num_ref_frames=4, num_ref_idx_active=3, mmco=True, reorder=True, gaps=1, poc_type=0,
                                poc_pattern=[0, 2, 1], slices_per_pic=2, non_ref_every=4, idr_period=13, wmb=5, hmb=4),
    "redundant_slices":    dict(n_pics=12, seed=32, wmb=6, hmb=4, slices_per_pic=3, redundant=True, p_pcm=0.15, num_ref_frames=2,
                                num_ref_idx_active=2, idc=(0, 2), aso=True),
    "redundant_fmo":       dict(n_pics=10, seed=33, wmb=6, hmb=4, slices_per_pic=2, redundant=True, fmo=dict(type=1, groups=3), reorder=True,
                                num_ref_frames=3, num_ref_idx_active=2),
    "high_qp":             dict(n_pics=10, seed=21, wmb=6, hmb=4, max_qp=51, chroma_qp_offset=12, slices_per_pic=2, idc=(0, 2)),
    # level_prefix escapes (levels up to +-900 at QP 0..6) and the low-QP dequantisation branches
    "huge_levels_low_qp":  dict(n_pics=10, seed=50, wmb=5, hmb=4, huge_levels=True, min_qp=0, max_qp=6, slices_per_pic=2,
                                num_ref_frames=2, num_ref_idx_active=2),
    # the largest picture level 5.1 allows (36,864 macroblocks): the per-picture kernels must shed wavefronts to fit their
    # per-macroblock scheduling state into the CU's LDS
    "max_frame_size_4096x2304": dict(n_pics=2, seed=40, wmb=256, hmb=144, level=51, slices_per_pic=2, idc=(0,), p_skip=0.8),
    "vga_multi_slice":     dict(n_pics=4, seed=22, wmb=40, hmb=30, slices_per_pic=3, num_ref_frames=2, num_ref_idx_active=2, idc=(0, 2)),
}
for _s in range(160, 200):
    CONFIGS[f"random_{_s}"] = random_config(_s)


# ---- damaged streams: (writer keywords, damage keywords).  Slice NAL units are dropped or cut
# short; the decoder must report the same errors, conceal the same macroblocks the same way and keep the same DPB
# state as the reference.  (This first set predates the bit-flip work and keeps its golden answers: no frame_num gaps,
# no redundant slices, no flipped bits.  FLIPPED and OVERFLOW below add them.)
def _damaged(seed):
    cfg = random_config(seed)
    cfg["gaps"] = 0
    cfg["redundant"] = False
    return cfg, dict(seed=seed, p_drop=0.2, p_flip=0.0, p_trunc=0.2)


DAMAGED = {f"damaged_{_s}": _damaged(_s) for _s in range(0, 48)}

# ---- bit errors INSIDE slices (round 2).  One flipped bit per hit slice NAL unit, next to dropped and truncated ones:
# the parser must accept exactly what the reference's CAVLC accepts (src/h264bsd_cavlc.c:749-916, including the
# 15-coefficient block that runs one element past its end), fail at the same macroblock, roll back the same
# macroblocks (src/h264bsd_slice_data.c:298-354, with its off-by-one start) and reproduce what the frame buffer then
# shows.  frame_num gaps stay in; redundant slices are still left out ("known deviations").
# Some damaged streams make the reference OUTPUT memory it never initialised (a macroblock it counts as decoded but
# never wrote, in a frame buffer used for the first time): tests/golden/make_synth_golden.py finds those by decoding
# with two different heap fill bytes and lists them in tests/golden/reference_undefined.json; their golden answers (like
# all others) come from the reference run with its allocations starting out zeroed (tests/synth.py decode_reference).
def _flipped(seed):
    cfg = random_config(seed)
    cfg["redundant"] = False
    return cfg, dict(seed=seed, p_drop=0.1, p_flip=0.3, p_trunc=0.1)


FLIPPED = {f"flipped_{_s}": _flipped(_s) for _s in range(300, 364)}


# ---- residual outside [-512, 511]: the only decode error the reference finds after dequantisation
# (src/h264bsd_transform.c:184-188).  The writer plants levels on both sides of the limit; the stream itself is intact.
def _overflow(seed):
    cfg = random_config(seed)
    cfg["redundant"] = False
    cfg["overflow"] = 0.02
    cfg["max_qp"] = max(cfg["max_qp"], 40)
    return cfg, dict(seed=seed, p_drop=0.0, p_flip=0.0, p_trunc=0.0)


OVERFLOW = {f"overflow_{_s}": _overflow(_s) for _s in range(400, 448)}

# ---- redundant slices in damaged pictures (round 2).  The reference decodes a redundant slice only while the primary
# picture is incomplete, keeps the PIXELS of the first decode of a macroblock and the METADATA of the last one
# (src/h264bsd_macroblock_layer.c:985-1046, the writes skipped at :1006 / :1110), restamps slice ids and slice-level
# filter parameters before it parses a macroblock (src/h264bsd_slice_data.c:140), and a redundant slice that fails
# un-decodes macroblocks the primary slice had decoded (:298-354).  Seeds whose random configuration has redundant slices;
# the last third also carries flipped bits.
def _redundant(seed, flip):
    cfg = random_config(seed)
    cfg["gaps"] = 0
    assert cfg["redundant"]
    return cfg, dict(seed=seed, p_drop=0.2, p_flip=flip, p_trunc=0.2)


_RED_SEEDS = [_s for _s in range(500, 900) if random_config(_s)["redundant"]][:48]
REDUNDANT = {f"redundant_{_s}": _redundant(_s, 0.3 if _i >= 32 else 0.0) for _i, _s in enumerate(_RED_SEEDS)}

# ---- streams that the large sweeps of round 2 found (tools/sweep.py): kept as regression fixtures.
# Heavy damage (30 % of the slice NAL units dropped, 30 % truncated, 50 % with a flipped bit, residuals near the range
# limit, frame_num gaps and redundant slices kept): a picture concealed without any valid slice that still writes the
# DPB's spare buffer, and slice parameters restamped over a macroblock that a failed redundant slice had un-decoded.
def _heavy(seed):
    cfg = random_config(seed)
    cfg["overflow"] = 0.03
    cfg["max_qp"] = max(cfg["max_qp"], 40)
    return cfg, dict(seed=seed, p_drop=0.3, p_flip=0.5, p_trunc=0.3)


SWEEP_FINDS = {f"heavy_{_s}": _heavy(_s) for _s in (300513, 300785, 301141, 305207, 800227)}
SWEEP_FINDS["redundant_flipped_106936"] = (random_config(106936), dict(seed=106936, p_drop=0.2, p_flip=0.2, p_trunc=0.2))
SWEEP_FINDS["redundant_flipped_505161"] = (random_config(505161), dict(seed=505161, p_drop=0.2, p_flip=0.2, p_trunc=0.2))
SWEEP_FINDS["redundant_flipped_5483"] = (dict(random_config(5483), gaps=0), dict(seed=5483, p_drop=0.2, p_flip=0.3, p_trunc=0.2))
SWEEP_FINDS["redundant_flipped_3053"] = (random_config(3053), dict(seed=3053, p_drop=0.2, p_flip=0.3, p_trunc=0.2))

# a bundled x264 stream (one slice per picture, 40x23 macroblocks) with slices cut short: concealment at picture scale
DAMAGED_BUNDLED = {"damaged_bundled_640x360": ("test_640x360", dict(seed=7, p_drop=0.04, p_flip=0.0, p_trunc=0.3)),
                   # ... and the full-size one (120x68 macroblocks) with flipped bits and cut slices: roll-back over more than ten
                   # macroblocks of a row (slice_data.c:318-333 counts max(width, 10)), concealment lists of thousands of entries
                   "damaged_bundled_1920x1080": ("test_1920x1080", dict(seed=8011, p_drop=0.03, p_flip=0.15, p_trunc=0.05))}
for _s in (956431, 953258):                     # P_8x8 re-decode failing on a missing reference: the failing quadrant's refAddr is already NULL
    SWEEP_FINDS[f"redundant_flipped_{_s}"] = (random_config(_s), dict(seed=_s, p_drop=0.1, p_flip=0.3, p_trunc=0.1))
for _s in (971573, 972160):                     # stale motion state from an earlier picture under a failed re-decode; three versions of one macroblock
    SWEEP_FINDS[f"redundant_flipped_{_s}"] = (dict(random_config(_s), gaps=0), dict(seed=_s, p_drop=0.05, p_flip=0.4, p_trunc=0.05))
# a rolled-back slice's pixels under a macroblock that is never written again, predicted from a macroblock the same
# (redundant) slice had only decoded again: the pre-pass job brings that one along in its first version
SWEEP_FINDS["redundant_flipped_1122884"] = (random_config(1122884), dict(seed=1122884, p_drop=0.02, p_flip=0.6, p_trunc=0.02))
# the macroblock counter says "complete" while a macroblock was never decoded (a slice failed before it wrote a record):
# the picture shows what the frame buffer held, and the record must say ABSENT — not what the reused job buffer held
SWEEP_FINDS["redundant_flipped_2005492"] = (random_config(2005492), dict(seed=2005492, p_drop=0.1, p_flip=0.4, p_trunc=0.1))
# ... and the deblocking filter treats that macroblock with the type, QP, coefficient counts and motion its mbStorage_t kept
# from an EARLIER picture and the slice parameters of the last slice that started on it (the one deviation left open at the
# end of round 2): a deblock-only record from the parser's persistent MbInfo
SWEEP_FINDS["redundant_flipped_2010846"] = (random_config(2010846), dict(seed=2010846, p_drop=0.1, p_flip=0.4, p_trunc=0.1))
