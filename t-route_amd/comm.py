"""Ranks of a multi-GPU routing job on one node, and the device plumbing of their hand-off -- all through the C ABI
(include/trmc.h, "communicator"): RCCL over xGMI when every rank has a device of its own, a shared-memory transport
when ranks share a device (a rehearsal on a one-GPU box) or there is no device at all (host-pointer collectives only).
No other GPU library is involved: device buffers, streams and events are the library's thin wrappers of the HIP
runtime, handled here as plain integers.

What ranks exchange is what the reference hands from one sub-network order to the next as
``flowveldepth_interorder`` (compute.py:882-897, consumed mc_reach.pyx:458-469): hydrographs of cut rows, and at the end
the outlet hydrographs.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib

ID_BYTES = 128


def _check(rc):
    _lib.check(rc)


# ---- device memory / streams / events ---------------------------------------------------------------------------------
class DeviceBuffer:
    """`nbytes` of zero-filled device memory on `device`; `.ptr` is the device address."""

    def __init__(self, device, nbytes):
        self.device, self.nbytes = int(device), int(nbytes)
        _lib.mark_hip_started()
        p = C.c_void_p(0)
        _check(_lib.lib().trmc_dev_alloc(self.device, self.nbytes, C.byref(p)))
        self.ptr = p.value or 0

    @classmethod
    def from_array(cls, device, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(device, arr.nbytes)
        if arr.nbytes:
            _check(_lib.lib().trmc_dev_upload(b.device, C.c_void_p(b.ptr), _lib.ptr(arr), arr.nbytes))
        return b

    def download(self, shape, dtype, stream=0, offset=0):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            _check(_lib.lib().trmc_dev_download(self.device, _lib.ptr(out), C.c_void_p(self.ptr + offset), out.nbytes,
                                                C.c_void_p(stream) if stream else None))
        return out

    def free(self):
        if getattr(self, "ptr", 0):
            _lib.lib().trmc_dev_free(self.device, C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceArray:
    """A block of results left in HBM: `.numpy()` copies it to the host (after the work queued on `stream`)."""

    def __init__(self, buf, shape, dtype, stream=0):
        self.buf, self.shape, self.dtype, self.stream = buf, tuple(shape), np.dtype(dtype), stream
        self.ptr = buf.ptr

    def numpy(self):
        return self.buf.download(self.shape, self.dtype, self.stream)

    def download_async(self, stream, out=None):
        """start the copy into a page-locked array on `stream`; complete after stream_synchronize(device, stream)"""
        if out is None:
            out = _lib.result_empty(self.shape, self.dtype, always_pinned=True)
        if out.nbytes:
            _check(_lib.lib().trmc_dev_download_async(self.buf.device, _lib.ptr(out), C.c_void_p(self.ptr), out.nbytes,
                                                      C.c_void_p(stream) if stream else None))
        return out


def stream_create(device, priority=None):
    """priority: None (ordinary, the runtime's default), -1 low, 0 ordinary, +1 high"""
    s = C.c_void_p(0)
    if priority is None:
        _check(_lib.lib().trmc_stream_create(int(device), C.byref(s)))
    else:
        _check(_lib.lib().trmc_stream_create_prio(int(device), int(priority), C.byref(s)))
    return s.value or 0


def stream_destroy(device, stream):
    if stream:
        _lib.lib().trmc_stream_destroy(int(device), C.c_void_p(stream))


def stream_synchronize(device, stream):
    _check(_lib.lib().trmc_stream_synchronize(int(device), C.c_void_p(stream) if stream else None))


def device_synchronize(device):
    _check(_lib.lib().trmc_device_synchronize(int(device)))


def event_create(device):
    e = C.c_void_p(0)
    _check(_lib.lib().trmc_event_create(int(device), C.byref(e)))
    return e.value or 0


def event_destroy(device, event):
    if event:
        _lib.lib().trmc_event_destroy(int(device), C.c_void_p(event))


def event_record(device, event, stream):
    _check(_lib.lib().trmc_event_record(int(device), C.c_void_p(event), C.c_void_p(stream) if stream else None))


def stream_wait_event(device, stream, event):
    _check(_lib.lib().trmc_stream_wait_event(int(device), C.c_void_p(stream) if stream else None, C.c_void_p(event)))


def device_copy(device, dst_ptr, src_ptr, nbytes, stream):
    _check(_lib.lib().trmc_dev_copy(int(device), C.c_void_p(dst_ptr), C.c_void_p(src_ptr), int(nbytes),
                                    C.c_void_p(stream) if stream else None))


def gather_rows(device, src_ptr, index_ptr, nrows, row_bytes, dst_ptr, stream):
    _check(_lib.lib().trmc_dev_gather_rows(int(device), C.c_void_p(src_ptr), C.c_void_p(index_ptr), int(nrows), int(row_bytes),
                                           C.c_void_p(dst_ptr), C.c_void_p(stream) if stream else None))


# ---- the communicator -----------------------------------------------------------------------------------------------------
def _rendezvous_dir():
    return os.environ.get("TRMC_COMM_DIR", "/tmp")


def exchange_id(rank, world, key, make_id):
    """Rank 0 makes the communicator id; the others receive it -- over a small shared-memory communicator of this launch
    (`trmc_comm_init_shm`: rank 0 creates the segment exclusively and every other rank only proceeds once the live rank 0 of
    THIS launch has answered its token), not through a file under the key: a file left by an earlier job with the same key
    is indistinguishable from a fresh one, and a stale id makes ncclCommInitRank hang."""
    if world == 1:
        return make_id()
    ctl = Comm(rank, world, -1, backend="shm", key=f"{key}_id", shm_bytes=1 << 16)
    try:
        mine = np.frombuffer(make_id() if rank == 0 else bytes(ID_BYTES), dtype=np.uint8)
        return bytes(ctl.all_gather_host(mine)[0])
    finally:
        ctl.close()


def default_key():
    """A name all ranks of one launch agree on and no other launch shares: TRMC_COMM_KEY when the launcher set one
    (bench.py does), else the launcher's pid (the ranks a multi-process launcher starts share their parent) and port."""
    k = os.environ.get("TRMC_COMM_KEY")
    if k:
        return k
    # (torchrun restarts the ranks of a failed launch under the same parent and port: its run id and restart count tell the
    # launches apart; a key that does repeat is still safe -- a segment is only ever used by the launch that created it)
    return (f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{os.environ.get('TORCHELASTIC_RUN_ID', '')}"
            f"_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}")


_PROBE = """
import sys
sys.path.insert(0, {root!r})
import numpy as np
from troute_amd import comm as X
c = X.Comm({rank}, {world}, {device}, backend="rccl", key={key!r})
send = X.DeviceBuffer.from_array({device}, np.full(256, {rank}, np.uint8))
recv = X.DeviceBuffer({device}, 256 * {world})
st = X.stream_create({device})
c.all_gather(send.ptr, recv.ptr, 256, st)
got = recv.download(({world}, 256), np.uint8, st)
assert (got == np.arange({world}, dtype=np.uint8)[:, None]).all()
c.barrier()
c.close()
"""


def probe_rccl(rank, world, device, key, timeout=None):
    """Can the ranks of this launch start RCCL and move a block through it?  Tried in a CHILD process per rank (the children
    form the same collective), under a time-out: a communicator that cannot start on this node (no peer access, a
    device shared by two ranks, a driver without the IPC mode RCCL needs) fails or hangs THERE, not in the job."""
    import subprocess
    import sys
    if timeout is None:
        timeout = float(os.environ.get("TRMC_COMM_PROBE_TIMEOUT_S", "90"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _PROBE.format(root=root, rank=int(rank), world=int(world), device=int(device), key=str(key))
    try:
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
        return r.returncode == 0
    except subprocess.TimeoutExpired:
        return False


class Comm:
    """One rank's handle.  backend: "rccl" | "shm" | "auto" | "probe".
    "auto": RCCL when every rank of the node has a device of its own AND a probe of it succeeds on every rank (probe_rccl;
    the verdicts are agreed over a shared-memory control communicator), else the shared-memory transport.  "probe": the
    same without counting devices (what the tests use to walk the fall-back on a one-GPU box)."""

    def __init__(self, rank, world, device, backend="auto", key=None, shm_bytes=0):
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        _lib.single_hw_queue_per_priority("troute_amd.comm.Comm")        # (before the runtime is first touched, if it can be)
        key = key or default_key()
        lib = _lib.lib()
        ndev = _lib.device_count()
        auto = backend in ("auto", "probe")
        if auto:
            want = world > 1 and self.device >= 0 and (backend == "probe" or ndev >= world)
            backend = "shm"
            if want and os.environ.get("TRMC_COMM_PROBE", "1") != "0":
                ctl = Comm(rank, world, -1, backend="shm", key=f"{key}_ctl", shm_bytes=1 << 16)
                ok = probe_rccl(rank, world, self.device, f"{key}_probe")
                agreed = bool(ctl.all_gather_host(np.array([1 if ok else 0], np.int32)).all())
                ctl.close()
                backend = "rccl" if agreed else "shm"
            elif want:
                backend = "rccl"
        h = C.c_void_p(0)
        if backend == "rccl":
            def make_id():
                buf = (C.c_char * ID_BYTES)()
                _check(lib.trmc_comm_unique_id(buf))
                return bytes(buf)
            try:
                blob = exchange_id(self.rank, self.world, key, make_id)
                _check(lib.trmc_comm_init(self.rank, self.world, blob, self.device, C.byref(h)))
            except Exception:
                # RCCL missing or unable to start on this node: with "auto" every rank falls back to the shared-memory
                # transport (the failure is the same on all of them); an explicit "rccl" is an error
                if not auto:
                    raise
                backend, h = "shm", C.c_void_p(0)
        self.backend = backend
        if backend == "rccl":
            pass
        elif backend == "shm":
            name = ("/trmc_" + "".join(ch if ch.isalnum() else "_" for ch in str(key)))[:120]
            _check(lib.trmc_comm_init_shm(self.rank, self.world, name.encode(), self.device, int(shm_bytes), C.byref(h)))
        else:
            raise ValueError("backend must be 'rccl', 'shm' or 'auto'")
        self._h = h
        self._key = key
        self.barrier()
        if backend == "rccl" and self.rank == 0:          # everybody has read the id
            try:
                os.remove(os.path.join(_rendezvous_dir(), f"trmc_comm_{key}.id"))
            except OSError:
                pass

    def all_gather(self, send_ptr, recv_ptr, nbytes, stream=0):
        """recv[world][nbytes] <- every rank's send[nbytes] (device addresses), ordered on `stream`"""
        _check(_lib.lib().trmc_comm_all_gather(self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(nbytes),
                                               C.c_void_p(stream) if stream else None))

    def all_gather_host(self, arr):
        """[world, *arr.shape] of every rank's (equally shaped) host array"""
        arr = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + arr.shape, dtype=arr.dtype)
        if arr.nbytes:
            _check(_lib.lib().trmc_comm_all_gather_host(self._h, _lib.ptr(arr), _lib.ptr(out), arr.nbytes))
        return out

    def all_gather_rows_host(self, arr):
        """list (one per rank) of host arrays whose FIRST dimension may differ from rank to rank"""
        arr = np.ascontiguousarray(arr)
        ns = self.all_gather_host(np.array([arr.shape[0]], dtype=np.int64))[:, 0]
        pad = np.zeros((int(ns.max()),) + arr.shape[1:], dtype=arr.dtype)
        pad[:arr.shape[0]] = arr
        allp = self.all_gather_host(pad)
        return [allp[r, :int(ns[r])] for r in range(self.world)]

    def all_reduce_max_host(self, arr):
        return self.all_gather_host(arr).max(axis=0)

    def barrier(self):
        _check(_lib.lib().trmc_comm_barrier(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().trmc_comm_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
