"""Calls every context-taking entry point of the C ABI with a NULL context and zero / NULL arguments: each must
return an error code (or VB_OK for the two documented no-ops) without touching the context.  Run as a script in a
child process by tests/test_cabi_null_ctx.py (a crash must not take pytest down) and by tests/run_asan.sh against the
host-sanitizer build."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viabel_amd import _lib  # noqa: E402

NOOP_OK = {'vb_destroy', 'vb_comm_destroy', 'vb_host_free', 'vb_dis_state_drop'}       # documented: NULL context (NULL block: free(NULL)) is a no-op
SKIP = {'vb_version', 'vb_device_count', 'vb_comm_unique_id', 'vb_last_error', 'vb_create', 'vb_legacy_rng_uid'}


def zero_of(argtype):
    if argtype in (ctypes.c_double,):
        return 0.0
    if argtype in (ctypes.c_int, ctypes.c_int64, ctypes.c_uint, ctypes.c_uint64, ctypes.c_size_t):
        return 0
    return None                                    # every pointer type


def main():
    lib = _lib.load()
    bad = []
    n = 0
    for name, (restype, argtypes) in sorted(_lib.SIGNATURES.items()):
        if name in SKIP or not argtypes or argtypes[0] is not _lib._ctx_p:
            continue
        rc = getattr(lib, name)(*[zero_of(t) for t in argtypes])
        n += 1
        if (rc == 0) != (name in NOOP_OK):
            bad.append((name, rc))
    # the context-free entry points: NULL outputs are rejected, a negative device is rejected
    cnt = ctypes.c_int(-1)
    assert lib.vb_device_count(None) != 0
    lib.vb_device_count(ctypes.byref(cnt))
    ctx = _lib._ctx_p()
    assert lib.vb_create(-1, ctypes.byref(ctx)) != 0 and not ctx.value
    assert lib.vb_create(0, None) != 0
    assert lib.vb_create(1 << 20, ctypes.byref(ctx)) != 0 and not ctx.value
    assert lib.vb_comm_unique_id(None) != 0
    assert isinstance(lib.vb_last_error(None), bytes) and lib.vb_version().startswith(b'viabel_hip')
    # the host-side generator (no context): NULL handles / outputs are rejected, destroy(NULL) is a no-op
    h = ctypes.c_void_p()
    assert lib.vb_legacy_rng_create(1, None) != 0
    assert lib.vb_legacy_rng_create(1, ctypes.byref(h)) == 0 and h.value
    for fn, args in (('vb_legacy_rng_randn', (None, None, 4, 0)), ('vb_legacy_rng_randn', (h, None, 4, 0)),
                     ('vb_legacy_rng_randn', (h, None, -1, 0)), ('vb_legacy_rng_standard_t', (h, 0.0, None, 0)),
                     ('vb_legacy_rng_chisquare', (None, 1.0, None, 1)), ('vb_legacy_rng_random_sample', (h, None, 2)),
                     ('vb_legacy_rng_get_state', (h, None, None, None, None)),
                     ('vb_legacy_rng_set_state', (h, None, 0, 0, 0.0))):
        if getattr(lib, fn)(*args) == 0:
            bad.append((fn, 0))
    assert lib.vb_legacy_rng_uid(None) == 0 and lib.vb_legacy_rng_uid(h) > 0
    lib.vb_legacy_rng_destroy(h)
    lib.vb_legacy_rng_destroy(None)
    if bad:
        print('entry points that did not reject a NULL context:', bad)
        return 1
    print('ok: %d entry points rejected a NULL context' % n)
    return 0


if __name__ == '__main__':
    sys.exit(main())
