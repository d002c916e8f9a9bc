"""CPU-only tests of the host side: hparams, state_dict layout, schedule, helpers, checkpoint round trip, the
C-ABI library (loads, exports every declared symbol, layout queries) and the loud-failure rule (no CPU fallback)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def fresh_hp(over=""):
    import hyperparams
    hp = hyperparams.hparams
    hp.override_from_dict(hyperparams.DEFAULTS)
    if over:
        hp.parse(over)
    return hp


def test_hparams_surface():
    hp = fresh_hp()
    assert hp.vocab_size == 6000 and hp.decoder_hidden == 768 and hp.adam_eps == 5e-8 and hp.multi_speaker is True
    hp.parse("transformer_dropout_rate=0.0,n_encoder_layer=2,multi_speaker=false,data_format=abc")
    assert hp.transformer_dropout_rate == 0.0 and hp.n_encoder_layer == 2 and hp.multi_speaker is False and hp.data_format == "abc"
    with pytest.raises(ValueError):
        hp.parse("not_a_param=1")
    with pytest.raises(ValueError):
        hp.parse("n_encoder_layer=abc")
    js = json.loads(hp.to_json())
    assert js["num_mels"] == 80 and "compute_dtype" in js and "vocab_size" in hp
    fresh_hp()


@pytest.mark.parametrize("tag", ["tiny", "tiny96", "default"])
def test_state_dict_layout_matches_reference(tag):
    from oracle import TINY, TINY96
    from transformer.tacotron import Tacotron
    hp = fresh_hp({"tiny": TINY, "tiny96": TINY96, "default": ""}[tag])
    m = Tacotron(hp)
    ref = json.load(open(os.path.join(G, "state_layout_%s.json" % tag)))
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref
    fresh_hp()


def test_lr_schedule_and_helpers_match_goldens():
    from transformer.tacotron import learning_rate_schedule
    from transformer import common
    hp = fresh_hp()
    g = dict(np.load(os.path.join(G, "g6_misc.npz")))
    for s, v in zip(g["lr_steps"], g["lr_values"]):
        assert learning_rate_schedule(int(s), hp) == pytest.approx(float(v), rel=1e-12)
    g1 = dict(np.load(os.path.join(G, "g1_helpers.npz")))
    assert np.array_equal(common.get_sinusoid_encoding_table(37, 64).numpy(), g1["pe_37_64"])
    assert np.array_equal(common.get_sinusoid_encoding_table(9, 7).numpy(), g1["pe_9_7"])
    assert np.array_equal(np.asarray(common.attention_bias(6, "causal")), g1["bias_causal_6"])
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], dtype=torch.bool)
    b = common.attention_bias(mask, "masking")
    assert np.array_equal(np.asarray(b), g1["bias_masking"]) and b.b2s_lengths.tolist() == [3, 5]
    with pytest.raises(ValueError):
        common.attention_bias(mask, "other")
    x3 = torch.arange(30, dtype=torch.float32).reshape(2, 5, 3) + 1
    lens = torch.tensor([3, 5])
    assert np.array_equal(common.impute(x3, lens).numpy(), g1["impute_cl"])
    loss = torch.arange(10, dtype=torch.float32).reshape(2, 5) * 0.25 + 1
    assert np.allclose(common.mask_reduce(loss, lens, True).numpy(), g1["mask_reduce_ps"])


def test_initialize_variables_distribution():
    from oracle import TINY
    from transformer.tacotron import Tacotron, initialize_variables
    hp = fresh_hp(TINY)
    torch.manual_seed(0)
    m = Tacotron(hp)
    initialize_variables(m)
    sd = m.state_dict()
    w = sd["decoder.decoder.ffn_layers.0.input_layer.weight"]
    std = float(np.sqrt(1.3 * 2.0 / ((w.shape[0] + w.shape[1]) / 2)))
    assert float(w.abs().max()) <= 2 * std + 1e-6 and abs(float(w.std()) - 0.88 * std) < 0.05 * std
    assert float(sd["encoder.speaker_layer.bias"].abs().max()) == 0.0
    assert abs(float(sd["encoder.embed.weight"].std()) - 1.0) < 0.05
    fresh_hp()


def test_text_and_checkpoint_roundtrip(tmp_path):
    from oracle import TINY
    from transformer.tacotron import Tacotron
    from utils import text, checkpoint, dict_send_to
    assert text.text_to_byte_sequence("hé") == [2, 104, 195, 169, 1]
    hp = fresh_hp(TINY)
    m = Tacotron(hp)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=hp.adam_eps)
    checkpoint.save_model(str(tmp_path), m, opt, None, 7)
    path = checkpoint.find_ckpt(str(tmp_path))
    assert path.endswith("model.ckpt-7")
    m2 = Tacotron(hp)
    wrapped = torch.nn.DataParallel(m2)               # `.module` prefix handling (utils/checkpoint.py:22-25,41-44)
    assert checkpoint.load_model(path, wrapped, None, None, "cpu") == 7
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert dict_send_to({"a": torch.ones(2), "names": ["x"]}, "cpu")["names"] == ["x"]
    # optimizer + scheduler round trip, highest step wins, 'module.'-prefixed model dicts are tolerated
    from functools import partial
    from transformer.tacotron import learning_rate_schedule
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=partial(learning_rate_schedule, hp=hp))
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step(); sched.step()
    checkpoint.save_model(str(tmp_path), m, opt, sched, 12)
    assert checkpoint.find_ckpt(str(tmp_path)).endswith("model.ckpt-12")
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3, eps=hp.adam_eps)
    sched2 = torch.optim.lr_scheduler.LambdaLR(opt2, lr_lambda=partial(learning_rate_schedule, hp=hp))
    assert checkpoint.load_model(checkpoint.find_ckpt(str(tmp_path)), m2, opt2, sched2, "cpu") == 12
    assert sched2.last_epoch == 1 and len(opt2.state_dict()["state"]) == len(list(m.parameters()))
    torch.save({"model": {"module." + k: v for k, v in m.state_dict().items()}}, str(tmp_path / "wrapped.pt"))
    m3 = Tacotron(hp)
    assert checkpoint.load_model(str(tmp_path / "wrapped.pt"), m3, None, None, "cpu") is None
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))
    assert checkpoint.find_ckpt(str(tmp_path / "nowhere")) is None
    fresh_hp()


def test_library_loads_and_exports_every_declared_symbol():
    from b2s_hip import lib
    l = lib.load()
    header = open(os.path.join(ROOT, "include", "b2s_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(b2s_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 35
    for name in sorted(declared):
        assert hasattr(l, name), "libb2s_hip.so does not export %s" % name
    assert declared == set(lib.EXPORTS), (declared ^ set(lib.EXPORTS))
    assert len(declared) <= 70, "C-ABI sprawl: %d entry points (scheduling variants belong in flags of ONE entry point)" % len(declared)
    assert l.b2s_version() >= 100
    # the dropout-site table is readable without a GPU, and unknown sites / decode-less sites are errors with a message
    op, kind, salt = C.c_uint32(), C.c_int(), C.c_int()
    lib.check(l.b2s_dropout_site(b"decoder.cross_attn", 3, 0, C.byref(op), C.byref(kind), C.byref(salt)))
    assert (op.value, kind.value, salt.value) == (2 * 4096 + 3 * 32 + 6, 1, 0)
    lib.check(l.b2s_dropout_site(b"decoder.ffn_res", 2, 1, C.byref(op), C.byref(kind), C.byref(salt)))
    assert (kind.value, salt.value) == (0, 1) and op.value != 2 * 4096 + 2 * 32 + 9
    assert l.b2s_dropout_site(b"postnet.conv", 0, 1, C.byref(op), None, None) != 0 and b"decode loop" in l.b2s_last_error()
    assert l.b2s_dropout_site(b"decoder.nothing", 0, 0, C.byref(op), None, None) != 0 and b"unknown dropout site" in l.b2s_last_error()


def test_product_library_has_no_lab_switches():
    """Measurement switches that change RESULTS (skip the encoder, drop the exchange ordering) exist only in -DB2S_LAB builds
    (csrc/build.sh --lab): the shipped library does not contain their names, so no environment variable can make it compute something
    else; and the number of environment switches the product reads at all stays small and documented."""
    from b2s_hip import lib
    blob = open(lib.LIB_PATH, "rb").read()
    assert b"B2S_LAB" not in blob
    names = set(re.findall(rb"B2S_[A-Z0-9_]{3,}", blob))
    header = open(os.path.join(ROOT, "include", "b2s_hip.h")).read()
    names -= {m.encode() for m in re.findall(r"#define\s+(B2S_[A-Z0-9_]+)", header)}      # flag names quoted in error messages
    allowed = {b"B2S_DW_GROUP", b"B2S_DX_BF16", b"B2S_ENC_FUSED", b"B2S_GEMM256_MIN_M", b"B2S_GEMM256_NB"}
    assert names <= allowed, sorted(names - allowed)
    src = os.path.join(ROOT, "few-shot-transformer-tts_amd")
    py = set()
    for d, _, fs in os.walk(src):
        for f in fs:
            if f.endswith(".py"):
                py |= set(re.findall(r"environ[^\n]*?[\"'](B2S_[A-Z0-9_]+)[\"']", open(os.path.join(d, f)).read()))
    assert py <= {"B2S_LIB_PATH", "B2S_FORCE_DP", "B2S_GRAD_PAYLOAD", "B2S_BN_BROADCAST", "B2S_DECODE_LANES", "B2S_DP_MODE", "B2S_COMPACT", "B2S_SIDE_STREAM", "B2S_DROPIN_OVERLAP"}, sorted(py)


def test_c_abi_layout_queries_and_errors():
    """Model creation / layout / workspace queries need no GPU; errors come back as codes + messages, never aborts."""
    from b2s_hip import lib
    from b2s_hip.engine import config_from_hparams
    l = lib.load()
    hp = fresh_hp()
    cfg = config_from_hparams(hp)
    h = lib.P()
    lib.check(l.b2s_model_create(C.byref(cfg), C.byref(h)))
    n = l.b2s_model_num_tensors(h)
    ref = json.load(open(os.path.join(G, "state_layout_default.json")))
    assert n == len(ref) == 177
    buf = C.create_string_buffer(256); shape = (C.c_int64 * 8)(); nd = C.c_int(); kind = C.c_int()
    for i in range(n):
        lib.check(l.b2s_model_tensor_info(h, i, buf, 256, shape, C.byref(nd), C.byref(kind)))
        assert [buf.value.decode(), [shape[k] for k in range(nd.value)]] == ref[i]
    assert l.b2s_encoder_ws_bytes(h, 14, 114) > 0 and l.b2s_decoder_ws_bytes(h, 14, 114, 582) > 1 << 28
    assert l.b2s_model_tensor_info(h, 999, buf, 256, shape, C.byref(nd), C.byref(kind)) != 0
    assert b"out of range" in l.b2s_last_error()
    # forward before binding -> error code, not a crash
    assert l.b2s_model_sync_weights(h, None, 0) != 0 and b"not bound" in l.b2s_last_error()
    l.b2s_model_destroy(h)
    bad = config_from_hparams(fresh_hp("decoder_hidden=512"))       # the reference crashes on this config too
    assert l.b2s_model_create(C.byref(bad), C.byref(h)) != 0 and b"decoder_hidden" in l.b2s_last_error()
    fresh_hp()


def test_no_cpu_fallback():
    """The HIP library is the only compute path: CPU tensors raise instead of silently running elsewhere."""
    from oracle import TINY
    from b2s_hip.lib import B2SError
    from transformer.tacotron import Tacotron
    from transformer.attention import MultiheadAttention
    hp = fresh_hp(TINY)
    m = Tacotron(hp)
    with pytest.raises(B2SError):
        m.encoder(torch.zeros(2, 5, dtype=torch.long), torch.tensor([5, 3]), torch.zeros(2, dtype=torch.long), torch.zeros(2, 8))
    with pytest.raises(B2SError):
        MultiheadAttention(64, 64, True, 2, 0.0)(torch.zeros(1, 4, 64), None, None)
    fresh_hp()


def test_stage_order_covers_all_parameters():
    from oracle import TINY
    from b2s_hip.engine import HipEngine
    from transformer.tacotron import Tacotron
    hp = fresh_hp(TINY)
    m = Tacotron(hp)
    eng = HipEngine(m, hp)                              # creating the handle needs no GPU
    stages = [eng.stage_of(n) for n, k in zip(eng.names, eng.kinds) if k == 1]
    assert min(stages) == 0 and max(stages) == eng.n_stages() - 1 and len(set(stages)) == eng.n_stages()
    assert eng.stage_of("postnet.conv_layers.0.weight") == 0
    assert eng.stage_of("decoder.decoder.ffn_layers.1.input_layer.weight") == 2
    assert eng.stage_of("decoder.decoder.self_attentions.0.qkv_transform.weight") == 3
    assert eng.stage_of("encoder.embed.weight") == eng.n_stages() - 1
    fresh_hp()


def test_gemm_tile_maps_are_bijections():
    """The block-id -> tile maps of the bf16 GEMM kernels (csrc/gemm_glds256.hip), restated in Python: the XCD-aware id map,
    the grouped-row-panel order of wide outputs and the per-problem XCD map of the grouped weight-gradient launch must each
    visit every tile exactly once, for tile counts that are not multiples of 8 / 4 too."""
    def xcd_tile_id(orig, nwg):
        xcd, q, r = orig & 7, nwg >> 3, nwg & 7
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (orig >> 3)

    def wide_order(rem, tiles_m, tiles_n, G=4):
        gsz = G * tiles_n
        grp, inn = divmod(rem, gsz)
        rows = min(G, tiles_m - grp * G)
        bx, r = divmod(inn, rows)
        return grp * G + r, bx

    for tiles_m, tiles_n in [(32, 24), (7, 16), (29, 32), (6, 16), (1, 17), (5, 64)]:
        n = tiles_m * tiles_n
        ids = sorted(xcd_tile_id(i, n) for i in range(n))
        assert ids == list(range(n))
        seen = {wide_order(xcd_tile_id(i, n), tiles_m, tiles_n) for i in range(n)}
        assert seen == {(by, bx) for by in range(tiles_m) for bx in range(tiles_n)}

    def grouped(i, tile0):
        p = 0
        while p + 1 < len(tile0) - 1 and i >= tile0[p + 1]:
            p += 1
        s0, cnt, xcd = tile0[p], tile0[p + 1] - tile0[p], i & 7
        before = 0
        for x in range(xcd):
            first = s0 + ((x - s0) & 7)
            before += (s0 + cnt - 1 - first) // 8 + 1 if first < s0 + cnt else 0
        return p, before + (i - (s0 + ((xcd - s0) & 7))) // 8

    for counts in [(72, 72, 54, 18, 18, 18), (36,), (24, 8, 32, 32), (1, 2, 3, 5, 7, 11, 13, 17), (9,), (18, 18, 18, 54, 72, 72, 36)]:
        tile0 = [0]
        for c in counts:
            tile0.append(tile0[-1] + c)
        got = sorted(grouped(i, tile0) for i in range(tile0[-1]))
        assert got == [(p, t) for p, c in enumerate(counts) for t in range(c)], counts
        # every XCD gets within one tile of an equal share of every problem
        for p, c in enumerate(counts):
            share = [sum(1 for i in range(tile0[p], tile0[p + 1]) if i % 8 == x) for x in range(8)]
            assert max(share) - min(share) <= 1


def test_grad_bucketer_split_and_prefix_wait():
    """GradBucketer.split: no bucket straddles the decoder | encoder boundary of the flat gradient buffer, and wait_prefix(limit) waits for
    exactly the all-reduces below it once they tile [0, limit) -- what lets the trainer update the decoder / postnet parameters while the
    encoder's gradients are still being produced and exchanged.  bf16 payload without consume_wire: the prefix is unpacked as well."""
    from b2s_hip.dp import GradBucketer

    class Work(object):
        def __init__(self, log, rng): self.log, self.rng = log, rng
        def wait(self): self.log.append(self.rng)

    class FakeDist(object):
        def __init__(self): self.waited = []; self.n = 0
        def all_reduce(self, t, group=None, async_op=False):
            self.n += 1
            t.mul_(2)
            return Work(self.waited, t.numel())
    flat = torch.arange(100, dtype=torch.float32)
    ranges = {0: (0, 30), 1: (30, 30), 2: (30, 60), 3: (60, 70), 4: (70, 100)}
    d = FakeDist()
    bk = GradBucketer(flat.clone(), ranges, 5, bucket_elems=50, dist=d, payload="bf16")
    bk.split = 60
    bk.begin_step()
    bk.stage_done(0)
    assert not bk.wait_prefix(60) and d.waited == []           # stage 2 has not reported: nothing waited for
    bk.stage_done(1); bk.stage_done(2)
    assert bk.launched == [(0, 60)]                             # closed AT the split although a 50-element bucket was already full at 60 only
    bk.stage_done(3)
    assert bk.launched == [(0, 60)]                             # (60, 70) is pending: below the bucket size
    assert bk.wait_prefix(60) and d.waited == [60]
    assert torch.equal(bk.flat[:60], (flat[:60].to(torch.bfloat16) * 2).float()) and torch.equal(bk.flat[60:], flat[60:])     # prefix unpacked only
    bk.stage_done(4)
    bk.finish()
    assert bk.launched == [(0, 60), (60, 100)] and d.waited == [60, 40]
    # without the split the same stages merge across the boundary
    bk2 = GradBucketer(flat.clone(), ranges, 5, bucket_elems=65, dist=FakeDist())
    bk2.begin_step()
    for s in range(5):
        bk2.stage_done(s)
    bk2.finish()
    assert bk2.launched == [(0, 70), (70, 100)]


def test_grad_bucketer_reports_missing_stage():
    """finish() verifies that the launched all-reduce ranges tile the flat gradient buffer: a backward stage that never
    reported (e.g. its hook raised) must not pass silently as 'reduced'."""
    from b2s_hip.dp import GradBucketer

    class Work(object):
        def wait(self): pass

    class FakeDist(object):
        def __init__(self): self.calls = []
        def all_reduce(self, t, group=None, async_op=False):
            self.calls.append(t.numel()); return Work()
    flat = torch.zeros(100)
    ranges = {0: (0, 30), 1: (30, 30), 2: (30, 70), 3: (70, 100)}
    d = FakeDist()
    bk = GradBucketer(flat, ranges, 4, bucket_elems=35, dist=d)
    bk.begin_step()
    for s in range(4):
        bk.stage_done(s)
    bk.finish()
    assert bk.launched == [(0, 70), (70, 100)] and d.calls == [70, 30]
    bk.begin_step()
    for s in (0, 1, 2):                                   # stage 3 never reports
        bk.stage_done(s)
    with pytest.raises(RuntimeError, match="did not report"):
        bk.finish()
    bk.begin_step()
    for s in (0, 1, 2):
        bk.stage_done(s)
    bk.finish(expect_all=False)                           # frozen encoder: the tail of the buffer is never exchanged


def test_dropout_hash_statistics():
    """The counter RNG behind every dropout mask (csrc/b2s_common.h: b2s_keep = lowbias32(idx * golden + key) >= p * 2^32, restated in
    oracle/rng.py and pinned bit for bit to the device by tests/test_gpu_ops.py): avalanche of the hash, keep-rate, independence of
    neighbouring elements, of rows and diagonals of an attention-shaped [rows, 582] mask, binomial row / column sums, independent ops."""
    import numpy as np
    from oracle import rng
    g = np.random.default_rng(1)
    x = g.integers(0, 1 << 26, 1 << 14, dtype=np.uint64)
    for key in (0x1234567, 0xdeadbeef):
        h0 = rng.rand32(x, key)
        for i in range(26):
            d = h0 ^ rng.rand32(x ^ np.uint64(1 << i), key)
            flips = np.array([((d >> np.uint64(o)) & np.uint64(1)).mean() for o in range(32)])
            assert flips.min() > 0.46 and flips.max() < 0.54, (key, i, flips.min(), flips.max())
    rows, Lk = 2048, 582
    for p, seed, op in ((0.1, 1234, 5), (0.5, 99, 77)):
        keep = rng.keep_mask(p, seed, op, rows * Lk).reshape(rows, Lk).astype(np.float64)
        assert abs((1 - keep.mean()) - p) < 4 * (p * (1 - p) / keep.size) ** 0.5 + 2e-5
        k = keep - keep.mean()
        var = k.var()
        corr = lambda a, b: abs(float((a * b).mean() / var))
        assert corr(k[:, :-1], k[:, 1:]) < 4e-3 and corr(k[:, :-2], k[:, 2:]) < 4e-3
        assert corr(k[:-1], k[1:]) < 4e-3 and corr(k[:-1, :-1], k[1:, 1:]) < 4e-3
        assert 0.9 < keep.sum(1).var() / (Lk * p * (1 - p)) < 1.1 and 0.85 < keep.sum(0).var() / (rows * p * (1 - p)) < 1.15
        other = rng.keep_mask(p, seed, op + 1, rows * Lk).reshape(rows, Lk)
        assert abs((keep.astype(bool) == other).mean() - ((1 - p) ** 2 + p ** 2)) < 3e-3
    # the training kernels' attention-weight masks (keep_mask_attn: a fully hashed seed per row, ONE mixing step + two multiplies per four keys,
    # 16-bit fields): the same properties, the drop rate quantised to 1 / 65536, the four fields of a quad independent of each other, odd key counts
    for p, seed, op, Lk in ((0.1, 1234, 8324, 582), (0.1, 77, 8358, 113), (0.5, 99, 77, 582)):
        keep = rng.keep_mask_attn(p, seed, op, rows, Lk).astype(np.float64)
        pq = (rng.drop_thresh(p) >> 16) / 65536.0
        assert abs(pq - p) < 2e-5
        assert abs((1 - keep.mean()) - pq) < 4 * (p * (1 - p) / keep.size) ** 0.5
        k = keep - keep.mean()
        var = k.var()
        corr = lambda a, b: abs(float((a * b).mean() / var))
        ev = Lk & ~3
        lim = max(4e-3, 4.5 / keep.size ** 0.5)                                       # (4.5 standard errors of an empirical correlation at the short key count)
        for i in range(4):                                                        # within a quad
            for j in range(i + 1, 4):
                assert corr(k[:, i:ev:4], k[:, j:ev:4]) < 2 * lim, (i, j)
        for d in (1, 2, 3, 4, 8, 16):
            assert corr(k[:, :-d], k[:, d:]) < lim, d
        assert corr(k[:-1], k[1:]) < lim and corr(k[:-1, :-1], k[1:, 1:]) < lim and corr(k[:-1, 1:], k[1:, :-1]) < lim
        assert 0.9 < keep.sum(1).var() / (Lk * p * (1 - p)) < 1.1 and 0.8 < keep.sum(0).var() / (rows * p * (1 - p)) < 1.2
        other = rng.keep_mask_attn(p, seed, op + 1, rows, Lk)
        assert abs((keep.astype(bool) == other).mean() - ((1 - p) ** 2 + p ** 2)) < 3e-3
        flat = rng.keep_mask(p, seed, op, rows * Lk).reshape(rows, Lk)              # and unrelated to the element rule of the same op
        assert abs((keep.astype(bool) == flat).mean() - ((1 - p) ** 2 + p ** 2)) < 3e-3
        assert np.array_equal(rng.keep_mask_attn(p, seed, op, 7, Lk, row0=100), keep[100:107].astype(bool))


def test_persistent_gemm_ticket_register_is_not_touched_between_draw_and_read():
    """csrc/gemm_glds256.hip draws its tickets with an asynchronous atomic (inline asm) whose result register is read two K steps later; that holds only
    while the compiler keeps the value in one register and writes nothing else to it in between.  tools/check_persist_isa.py compiles the file to ISA and
    checks it for every instantiation (about a minute)."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_persist_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
