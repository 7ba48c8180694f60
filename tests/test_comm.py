"""Host-C++ sharding over RCCL (include/scenelib2_amd_comm.h, scenelib2_amd/libscenelib2_amd_comm.so, examples/sharded_monoslam.cpp).

CPU: the library exports exactly what its header declares; the partition rule every scatter / gather uses (sl2_shard_range)
equals the Python launcher's (scenelib2_amd/sharding.py) for every (total, ranks) - the N > 1 arithmetic, with no collective
involved.  GPU (one device on the test box): a single-rank communicator, frames scattered from the root, three sequences
stepped, states and covariance blocks gathered - against the oracle fed the same bytes."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "scenelib2_amd", "libscenelib2_amd_comm.so")
HDR = os.path.join(ROOT, "include", "scenelib2_amd_comm.h")


def _declared():
    text = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(sl2_[a-z_0-9]+)\s*\(", text)))


def _exported():
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    return sorted(ln.split()[-1] for ln in out.splitlines() if " T sl2_" in ln)


def test_comm_library_exports_exactly_its_header():
    assert os.path.exists(LIB), "scenelib2_amd/libscenelib2_amd_comm.so not built (make -C scenelib2_amd/csrc)"
    assert _exported() == _declared()


def test_shard_range_is_the_block_partition_of_the_python_launcher():
    from scenelib2_amd import sharding
    L = C.CDLL(LIB)
    L.sl2_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for total in (0, 1, 7, 8, 1024, 8192, 4099):
        for world in (1, 2, 3, 8):
            covered = []
            for rank in range(world):
                f, n = C.c_int(-1), C.c_int(-1)
                assert L.sl2_shard_range(total, world, rank, C.byref(f), C.byref(n)) == 0
                assert (f.value, n.value) == sharding.shard_range(total, world, rank)
                covered += list(range(f.value, f.value + n.value))
            assert covered == list(range(total))                       # every sequence exactly once, in rank order
    f, n = C.c_int(0), C.c_int(0)
    assert L.sl2_shard_range(8, 2, 2, C.byref(f), C.byref(n)) != 0    # rank out of range
    L.sl2_gather_row_doubles.argtypes = [C.c_int, C.c_int]
    assert [L.sl2_gather_row_doubles(k, 32) for k in (0, 1, 2, 3)] == [13, 182, 13 + 96, -1]


@pytest.mark.gpu
def test_sharded_example_on_one_gpu_matches_the_oracle(tmp_path):
    """examples/sharded_monoslam --gpus 1 --per-gpu 3: ncclCommInitAll on one device, the root's frames to its own block
    through sl2_scatter_frames, three engines' worth of sequences in one batch, ncclAllGather of xv and Pxx.  (More ranks need
    more devices than the test box has: the N > 1 partition is the CPU test above, the collectives are RCCL's.)"""
    import oracle_api as oa
    from mapping_helpers import make_mapping_sequence, oracle_for
    from test_gpu_headless_example import _write_scene
    exe = os.path.join(ROOT, "examples", "sharded_monoslam")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=12)
    cfg, fd = _write_scene(str(tmp_path), cam, params, spec, frames, templates)
    dump = os.path.join(str(tmp_path), "states.txt")
    out = subprocess.run([exe, "--cfg", cfg, "--frames", fd, "--gpus", "1", "--per-gpu", "3", "--dump", dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = np.loadtxt(dump)
    assert rows.shape == (3, 13 + 169)
    s = oracle_for(cam, params, spec, templates, oa)
    for k in range(1, 13):
        s.go_one_step(frames[k], False, False)
    xv, Pxx = s.get_state()
    for b in range(3):
        assert np.abs(rows[b, :13] - xv).max() < 1e-12
        assert np.abs(rows[b, 13:].reshape(13, 13) - Pxx).max() <= 1e-11 * np.abs(Pxx).max()
    two = subprocess.run([exe, "--cfg", cfg, "--frames", fd, "--gpus", "2"], capture_output=True, text=True, timeout=120)
    assert two.returncode == 3 and "HIP device" in two.stderr          # fewer devices than ranks: refused, not mislabelled


def test_comm_entry_points_refuse_bad_arguments_and_a_missing_device():
    """Error behaviour without a GPU in reach of the call: status codes, never a crash (the reference has no error codes at all;
    the C ABI's convention is SL2_OK / SL2_ERR_*).  On a box without a HIP device the constructors answer SL2_ERR_NO_DEVICE."""
    from scenelib2_amd import _lib
    L = C.CDLL(LIB)
    L.sl2_comm_last_error.restype = C.c_char_p
    out = (C.c_void_p * 2)()
    assert L.sl2_comm_create_all(0, None, out) == 1 and b"bad argument" in L.sl2_comm_last_error()          # SL2_ERR_INVALID
    assert L.sl2_comm_create(None, 1, 0, 0, out) == 1
    ident = (C.c_ubyte * 128)()
    assert L.sl2_comm_create(ident, 2, 2, 0, out) == 1                                                       # rank out of range
    assert L.sl2_scatter_frames(None, 0, None, C.c_size_t(1), 1, None, None) == 1
    assert L.sl2_gather_states(None, None, 0, None, None) == 1
    assert L.sl2_comm_rank(None) == -1 and L.sl2_comm_nranks(None) == 0 and L.sl2_comm_device(None) == -1
    L.sl2_comm_destroy(None)                                                                                 # a no-op
    if _lib.device_count() == 0:
        assert L.sl2_comm_create_all(1, None, out) == 4 and b"no HIP device" in L.sl2_comm_last_error()      # SL2_ERR_NO_DEVICE
        assert L.sl2_comm_create(ident, 1, 0, 0, out) == 4
