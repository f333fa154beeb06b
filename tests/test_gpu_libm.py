"""GPU leg of the libm pin: the device functions the kernel inlines (csrc/pt_device_math.h gl_cosf / gl_log10f / gl_powf,
+ sinf / expf / logf for the wgpu-shader mode) return, for EVERY binary32 argument of the domains the path uses, the
bits of the CPU restatement (oracle/glibc_flt32.h) -- which tests/test_libm_pin.py proves equal to glibc 2.35's libm
(reference call sites: src/math.rs:15-19, src/math/vec3.rs:80-90).  So kernel == oracle == what the Rust binary's libm
returns, argument by argument.  Runs through libmipt_diag.so (include/mipt_diag.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNK = 1 << 26      # 64 Mi arguments = 256 MiB of results per call


def _bits(x):
    return int(np.float32(x).view(np.uint32))


def _sweep(rrt, orc, op, first, last, y=0.0, ref_op=None):
    diag = rrt.load_diag()
    bad = 0
    examples = []
    b = first
    while b <= last:
        n = min(CHUNK, last - b + 1)
        dev = np.empty(n, dtype=np.float32)
        assert diag.mipt_debug_eval_range(op, b, n, float(y), dev.ctypes.data) == 0, diag.mipt_diag_last_error()
        x = np.arange(b, b + n, dtype=np.uint64).astype(np.uint32).view(np.float32)
        rop = op if ref_op is None else ref_op
        ref = orc.eval_array(rop, x, None if rop != 2 else np.float32(y), libm=orc.LIBM_GLIBC235, threads=16)
        d, r = dev.view(np.uint32), ref.view(np.uint32)
        neq = (d != r) & ~(np.isnan(dev) & np.isnan(ref))
        k = int(np.count_nonzero(neq))
        if k:
            bad += k
            for i in np.flatnonzero(neq)[:4]:
                examples.append((hex(b + int(i)), hex(int(d[i])), hex(int(r[i]))))
        b += n
    return bad, examples


def test_device_cosf_log10f_powf_equal_the_restatement_on_every_argument_of_the_path(rrt, orc):
    top = _bits(np.float32(6.283185) * np.float32(1.0))
    assert _sweep(rrt, orc, 0, 0, top) == (0, [])                                   # cos(theta), theta = 6.283185 * r
    assert _sweep(rrt, orc, 1, 0, _bits(1.0)) == (0, [])                            # log10(r), r in [0, 1]
    y = np.float32(1.0) / np.float32(2.4)
    assert _sweep(rrt, orc, 2, 0, _bits(1024.0), y) == (0, [])                      # powf(c, 1/2.4), c in [0, 2^10]


def test_device_functions_on_special_and_out_of_domain_arguments(rrt, orc):
    y = float(np.float32(1.0) / np.float32(2.4))
    for first, last in ((0x00000000, 0x00800fff), (0x3f7ff000, 0x3f801000), (0x42ef0000, 0x42f10000), (0x4a000000, 0x4a010000),
                        (0x7f7ff000, 0x80800fff), (0xbf7ff000, 0xbf801000), (0xc2ef0000, 0xc2f10000), (0xff7ff000, 0xffffffff)):
        for op, yy in ((0, 0), (16, 0), (18, 0), (1, 0), (17, 0), (2, y), (2, 2.2), (2, -3.0), (2, 0.0)):
            assert _sweep(rrt, orc, op, first, last, yy) == (0, []), (op, yy, hex(first))


def test_device_sinf_expf_powf22_on_the_wgpu_shading_domains(rrt, orc):
    # mode 1 (rt_compute.wgsl): sin/cos of phi = 2*pi*u and of theta <= pi/2 + ..., exp(-(1-c)*d) (<= 0), pow(texel, 2.2) on [0, 1]
    assert _sweep(rrt, orc, 16, 0, _bits(6.2831855)) == (0, [])
    assert _sweep(rrt, orc, 17, _bits(-0.0), _bits(-128.0)) == (0, [])
    assert _sweep(rrt, orc, 2, 0, _bits(1.0), 2.2) == (0, [])


def test_the_kernels_specialised_log10f_and_cosf_equal_the_general_functions_on_their_whole_domain(rrt, orc):
    """The scatter step (math.rs:15-19) calls gl_log10f_unit / gl_cosf_2pi -- straight-line forms valid on what rand_f32 can
    produce: log10 of {0} u [2^-32, 1], cos of 6.283185f * r in [0, 6.2831855].  EVERY binary32 of both domains must give the
    bits of the CPU restatement of glibc's log10f / cosf (ops 19 / 20 of the probe vs ops 1 / 0 of the oracle)."""
    top = _bits(np.float32(6.283185) * np.float32(1.0))
    assert _sweep(rrt, orc, 20, 0, top, ref_op=0) == (0, [])
    assert _sweep(rrt, orc, 19, 0, 0, ref_op=1) == (0, [])                             # log10(0) = -inf
    assert _sweep(rrt, orc, 19, _bits(2.0 ** -32), _bits(1.0), ref_op=1) == (0, [])


def test_floor_mod_of_the_bilinear_sampler(rrt):
    """Shading mode 1's texture wrap (rt_compute.wgsl textureSampleLevel, repeat addressing): floor_mod(i, W) == i mod W in the
    floored sense for |i| < 2^30 and every W in [1, 2^32), on edge cases and two million random pairs."""
    diag = rrt.load_diag()
    rng = np.random.default_rng(5)
    edge_i = np.array([0, 1, -1, 2, -2, 2 ** 30 - 1, -(2 ** 30 - 1), 999_999_999, -999_999_999, 12345, -12345], dtype=np.int64)
    edge_w = np.array([1, 2, 3, 7, 255, 256, 1000, 65535, 65536, 2 ** 30 - 1, 2 ** 30, 2 ** 30 + 1, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1], dtype=np.int64)
    ii, ww = np.meshgrid(edge_i, edge_w)
    i = np.concatenate([ii.ravel(), rng.integers(-(2 ** 30) + 1, 2 ** 30, 2_000_000)])
    w = np.concatenate([ww.ravel(), np.where(rng.random(2_000_000) < 0.5, rng.integers(1, 5000, 2_000_000), rng.integers(1, 2 ** 32, 2_000_000))])
    a = i.astype(np.int32).view(np.float32).copy()
    bb = w.astype(np.uint32).view(np.float32).copy()
    out = np.zeros(a.size, dtype=np.float32)
    assert diag.mipt_debug_eval(23, a.ctypes.data, bb.ctypes.data, a.size, out.ctypes.data) == 0
    assert np.array_equal(out.view(np.uint32).astype(np.int64), np.mod(i, w))
