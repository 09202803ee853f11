"""The exchange step of a sharded cube fit: one process per GPU, RCCL over xGMI through the
engine's C ABI (``nfa_comm_*``, csrc/nfa_comm.h; librccl.so is dlopen'ed by the engine).

The reference's processes exchange nothing while sampling and meet through chunk files
(nestfit/main.py:516-523, docs/store_spec.rst:12-32); here the ranks all-gather fixed-size
per-pixel records once at the end and reduce a few doubles (timing, counts).  Any object with
``rank``, ``world``, ``allgather(x)``, ``allreduce(x, op)`` and ``barrier()`` serves as a
communicator for `gather_pixel_records`: `RcclComm` on GPUs, a gloo-backed stand-in in the CPU
tests (tests/comm_gloo.py), `SoloComm` for one process.
"""
import ctypes as C
import os
import socket
import struct
import time

import numpy as np

from . import _ffi

_OPS = {'sum': 0, 'max': 2, 'min': 3}          # ncclRedOp_t (rccl.h:448-451)


class SoloComm:
    """world = 1: every collective is the identity."""
    rank, world = 0, 1

    def allgather(self, x):
        return np.ascontiguousarray(x, dtype=np.float64).ravel().copy()

    def allreduce(self, x, op='sum'):
        return np.ascontiguousarray(x, dtype=np.float64).copy()

    def barrier(self):
        pass

    def close(self):
        pass


def _exchange_id(rank, world, make_id, addr, port, timeout=180.0):
    """Rank 0 creates the 128-byte ncclUniqueId and hands it to every other rank over a loopback /
    LAN TCP socket (the launcher's MASTER_ADDR, MASTER_PORT + 1: the port itself belongs to the
    launcher's own store)."""
    if rank == 0:
        uid = make_id()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind(('', port))
        srv.listen(world)
        srv.settimeout(timeout)
        try:
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(struct.pack('<I', len(uid)) + uid)
        finally:
            srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                n = struct.unpack('<I', _recv_exact(s, 4))[0]
                return _recv_exact(s, n)
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)


def _recv_exact(sock, n):
    buf = b''
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise OSError('connection closed during the unique-id exchange')
        buf += part
    return buf


class RcclComm:
    """RCCL communicator of this process's GPU (call ``nestfit_amd.set_device(local_rank)`` first)."""

    def __init__(self, rank, world, unique_id):
        self.rank, self.world = int(rank), int(world)
        self._h = C.c_void_p()
        uid = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        _ffi.check(_ffi.load().nfa_comm_create(C.byref(self._h), uid, self.rank, self.world))

    @classmethod
    def from_env(cls, port_offset=1):
        """Under ``python -m torch.distributed.run`` (or any launcher that sets RANK, WORLD_SIZE,
        MASTER_ADDR, MASTER_PORT)."""
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = int(os.environ.get('MASTER_PORT', '29500')) + port_offset

        def make_id():
            buf = (C.c_ubyte * 128)()
            _ffi.check(_ffi.load().nfa_comm_unique_id(buf))
            return bytes(buf)

        return cls(rank, world, _exchange_id(rank, world, make_id, addr, port))

    def allgather(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        out = np.empty(x.size * self.world)
        _ffi.check(_ffi.load().nfa_comm_allgather(self._h, _ffi.dptr(x), x.size, _ffi.dptr(out)))
        return out

    def allreduce(self, x, op='sum'):
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        _ffi.check(_ffi.load().nfa_comm_allreduce(self._h, _ffi.dptr(x), x.size, _OPS[op]))
        return x

    def barrier(self):
        _ffi.check(_ffi.load().nfa_comm_barrier(self._h))

    def close(self):
        if self._h:
            _ffi.load().nfa_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TcpComm:
    """The same interface over plain TCP sockets, rank 0 as the hub: for ranks that share one GPU
    (RCCL refuses two ranks on a device: rehearsals of the N > 1 path on a one-GPU box) or have none.
    Latency-bound traffic of a few KB; not a data path."""

    def __init__(self, rank, world, addr='127.0.0.1', port=29517, timeout=180.0):
        self.rank, self.world = int(rank), int(world)
        self._peers = []
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('', port))
            srv.listen(world)
            srv.settimeout(timeout)
            peers = {}
            for _ in range(world - 1):
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                peers[struct.unpack('<I', _recv_exact(conn, 4))[0]] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    self._hub = socket.create_connection((addr, port), timeout=timeout)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            self._hub.sendall(struct.pack('<I', self.rank))

    @classmethod
    def from_env(cls, port_offset=2):
        return cls(int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), os.environ.get('MASTER_ADDR', '127.0.0.1'),
                   int(os.environ.get('MASTER_PORT', '29500')) + port_offset)

    def allgather(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        if self.world == 1:
            return x.copy()
        if self.rank == 0:
            parts = [x] + [np.frombuffer(_recv_exact(c, 8 * x.size), dtype=np.float64) for c in self._peers]
            out = np.concatenate(parts)
            for c in self._peers:
                c.sendall(out.tobytes())
            return out
        self._hub.sendall(x.tobytes())
        return np.frombuffer(_recv_exact(self._hub, 8 * x.size * self.world), dtype=np.float64).copy()

    def allreduce(self, x, op='sum'):
        x = np.ascontiguousarray(x, dtype=np.float64)
        allx = self.allgather(x).reshape(self.world, -1)
        return {'sum': allx.sum(axis=0), 'max': allx.max(axis=0), 'min': allx.min(axis=0)}[op].reshape(x.shape)

    def barrier(self):
        self.allgather(np.zeros(1))

    def close(self):
        for c in self._peers:
            c.close()
        self._peers = []
        if self.rank != 0 and getattr(self, '_hub', None) is not None:
            self._hub.close()
            self._hub = None


def comm_from_env(rccl_timeout=90.0):
    """The communicator of a rank started by a launcher (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT): RCCL over
    xGMI when every rank gets its RCCL communicator up, the socket communicator otherwise -- decided together, so
    that a node where RCCL cannot start (a missing library, a bootstrap that finds no interface) still runs the
    sharded job: the data path has no collective, only the barrier / max-time / record gather travel here.
    Returns (communicator, 'rccl' | 'tcp' | 'solo')."""
    import threading
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return SoloComm(), 'solo'
    rank = int(os.environ['RANK'])
    tcp = TcpComm.from_env()
    # rank 0's ncclUniqueId travels over the socket communicator (128 bytes as 16 doubles, bit for bit)
    uid = np.zeros(17)
    if rank == 0:
        try:
            buf = (C.c_ubyte * 128)()
            _ffi.check(_ffi.load().nfa_comm_unique_id(buf))
            uid[:16] = np.frombuffer(bytes(buf), dtype=np.float64)
            uid[16] = 1.0
        except Exception as exc:                       # no RCCL on this node: everybody takes the sockets
            print(f'nestfit_amd.comm: no RCCL unique id ({exc}); using the socket communicator', flush=True)
    uid = tcp.allgather(uid).reshape(world, 17)[0]
    state = {}
    t = None

    def attempt():
        try:
            state['comm'] = RcclComm(rank, world, uid[:16].tobytes())
        except Exception as exc:
            state['error'] = str(exc)
    if uid[16] == 1.0:
        t = threading.Thread(target=attempt, daemon=True)
        t.start()
        t.join(rccl_timeout)
    mine = 1.0 if 'comm' in state else 0.0
    everyone = tcp.allreduce(np.array([mine]), 'min')[0]
    if everyone == 1.0:
        tcp.close()
        return state['comm'], 'rccl'
    if rank == 0:
        print(f'nestfit_amd.comm: RCCL communicator not available on every rank '
              f'({state.get("error", "timed out" if uid[16] == 1.0 else "no unique id")}); using the socket communicator', flush=True)
    # a rendezvous that timed out may still sit in its (daemon) thread: callers that are about to exit can look at
    # `stuck_thread` and leave through os._exit instead of waiting for the library's teardown
    tcp.stuck_thread = t if (uid[16] == 1.0 and t.is_alive()) else None
    return tcp, 'tcp'


def gather_pixel_records(records, comm):
    """All-gather fixed-size per-pixel result records (float64 [n_local, width]) from every rank;
    returns the concatenation in rank order.  Ranks may own different numbers of pixels (stripes of
    a cube whose width is not a multiple of the world size): counts go first, the records padded to
    the largest count second."""
    records = np.ascontiguousarray(records, dtype=np.float64)
    assert records.ndim == 2
    width = records.shape[1]
    counts = comm.allgather(np.array([records.shape[0]], dtype=np.float64)).astype(np.int64)
    n_max = int(counts.max())
    if n_max == 0:
        return np.zeros((0, width))
    padded = np.zeros((n_max, width))
    padded[:records.shape[0]] = records
    allrec = comm.allgather(padded).reshape(comm.world, n_max, width)
    return np.concatenate([allrec[r, :counts[r]] for r in range(comm.world)], axis=0)
