# coding=utf-8
"""How the ranks of `duet --gpus N` talk to each other without torch (round 4).

The product's data-path exchange is ONE all-gather of fixed-size record blocks.  Round 3 did it
through torch.distributed, which made every rank import torch and wait for its rendezvous: about a second before any work
started.  Now a rank needs numpy and the native libraries only:

  * `TcpStar` -- a star of TCP connections on MASTER_ADDR:MASTER_PORT (rank 0 listens, the environment is what
    duet_amd/launch.py or torchrun sets): broadcast of a small blob from rank 0 and a host-side all-gather.  It carries the
    RCCL unique id, and in the one-GPU plumbing mode (DUET_ONE_GPU=1: every rank on device 0, where RCCL cannot run two ranks)
    the record blocks themselves.
  * `RcclGather` -- the collective inside libduet_ef.so (duet_comm_*): ncclCommInitRank from rank 0's id, ncclAllGather over
    xGMI on device buffers of the rank's context.  `ef_allgather` is a rank's whole data path in ONE library call
    (duet_comm_ef_allgather): the shard's arrays up, the three kernels writing straight into the rank's record block on the
    device, the rows-kept-per-CHROM-text counters and the status word appended by a small kernel, ncclAllGather on the
    kernels' stream, one copy of the gathered blocks back -- the results never visit the host before the collective.
  * `HostGather` -- the same interface over the TCP star alone.

Every blocking step is bounded by `timeout` seconds (DUET_RDZV_TIMEOUT): the TCP star's accepts, reads and writes here, and
inside the library ncclCommInitRank (run on a helper thread, waited for with a deadline) and the wait behind the all-gather
(the stream is polled against a deadline) -- DUET_ERR_TIMEOUT -> `CommTimeout`.  A rank that never arrives makes the others
fail, not hang; a rank that gave up on RCCL must leave through os._exit (duet_amd/multi.py does): a helper thread may still
sit inside RCCL."""

import os
import socket
import struct
import time

import numpy as np


class CommError(RuntimeError):
    pass


class CommTimeout(CommError):
    """A collective step ran into DUET_RDZV_TIMEOUT; the process should end itself with os._exit."""


def block_bytes(n_max, n_slots):
    """Bytes of one rank's record block: ps u32[n_max] | pred u8[n_max] | pad to 16 | status u32 + 12 | kept u64[n_slots]
    (include/duet_ef.h: duet_comm_block_bytes)."""
    return (5 * int(n_max) + 15) // 16 * 16 + 16 + 8 * int(n_slots)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise CommError('peer closed the connection')
        got += k
    return bytes(buf)


def _send_blob(sock, data):
    sock.sendall(struct.pack('<Q', len(data)))
    sock.sendall(data)


def _recv_blob(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class TcpStar(object):
    def __init__(self, rank, world, addr=None, port=None, timeout=300.0):
        self.rank, self.world = int(rank), int(world)
        self.timeout = float(timeout)
        self.peers = {}
        self.sock = None
        if self.world == 1:
            return
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = int(port if port is not None else os.environ['MASTER_PORT'])
        deadline = time.time() + self.timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            try:
                while len(self.peers) < self.world - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise CommError('rendezvous: %d of %d ranks arrived within %.0f s' % (len(self.peers) + 1, self.world, self.timeout))
                    srv.settimeout(left)
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        continue
                    # the 4-byte rank id is read under what is LEFT of the deadline; an id that is not an expected peer's
                    # (a stray local connection to the port, a duplicate) is turned away instead of derailing the rendezvous
                    conn.settimeout(max(0.05, deadline - time.time()))
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    try:
                        (r,) = struct.unpack('<I', _recv_exact(conn, 4))
                    except (socket.timeout, CommError, OSError):
                        conn.close()
                        continue
                    if not (1 <= r < self.world) or r in self.peers:
                        conn.close()
                        continue
                    conn.settimeout(self.timeout)
                    self.peers[r] = conn
            except BaseException:
                for c in self.peers.values():
                    c.close()
                self.peers = {}
                raise
            finally:
                srv.close()
        else:
            while True:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                try:
                    s.settimeout(max(0.05, min(2.0, deadline - time.time())))
                    s.connect((addr, port))
                    break
                except (ConnectionRefusedError, socket.timeout, OSError):
                    s.close()
                    if time.time() > deadline:
                        raise CommError('rendezvous: rank 0 did not answer on %s:%d within %.0f s' % (addr, port, self.timeout))
                    time.sleep(0.01)
            s.settimeout(self.timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack('<I', self.rank))
            self.sock = s

    def bcast(self, data=None):
        """rank 0's bytes -> every rank"""
        if self.world == 1:
            return data
        try:
            if self.rank == 0:
                for r in sorted(self.peers):
                    _send_blob(self.peers[r], data)
                return data
            return _recv_blob(self.sock)
        except socket.timeout:
            raise CommError('broadcast timed out after %.0f s' % self.timeout)

    def allgather(self, data):
        """every rank's bytes -> list in rank order, on every rank"""
        if self.world == 1:
            return [data]
        try:
            if self.rank == 0:
                parts = [data] + [None] * (self.world - 1)
                for r in sorted(self.peers):
                    parts[r] = _recv_blob(self.peers[r])
                packed = b''.join(struct.pack('<Q', len(p)) + p for p in parts)
                for r in sorted(self.peers):
                    _send_blob(self.peers[r], packed)
                return parts
            _send_blob(self.sock, data)
            packed = _recv_blob(self.sock)
        except socket.timeout:
            raise CommError('all-gather timed out after %.0f s' % self.timeout)
        parts, at = [], 0
        try:
            for _ in range(self.world):
                (n,) = struct.unpack_from('<Q', packed, at)
                if at + 8 + n > len(packed):
                    raise struct.error('short')
                parts.append(packed[at + 8:at + 8 + n])
                at += 8 + n
        except struct.error:
            raise CommError('all-gather: rank 0 sent %d bytes that do not hold %d blocks' % (len(packed), self.world))
        return parts

    def close(self):
        for c in self.peers.values():
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = {}, None


class HostGather(object):
    """blocks (numpy uint8, the same size on every rank) -> [world, size] over the TCP star"""
    name = 'tcp'

    def __init__(self, star):
        self.star = star

    def allgather(self, block):
        block = np.ascontiguousarray(block, dtype=np.uint8)
        parts = self.star.allgather(block.tobytes())
        if any(len(p) != block.size for p in parts):
            raise CommError('all-gather: blocks of different sizes')
        return np.frombuffer(b''.join(parts), dtype=np.uint8).reshape(self.star.world, block.size).copy()

    def ef_allgather(self, soa, svlen_thres, suppread_thres, cand_slot, n_slots, n_max, compute):
        """A rank's data path with the star as the collective (the one-GPU plumbing mode, the CPU tests): `compute` is the
        rank's E/F (shard -> (block bytes ps | pred [| anything], status)); the trailer -- status word, rows kept per
        CHROM-text slot -- is assembled here.  -> uint8 [world, block_bytes(n_max, n_slots)]"""
        blk, status = compute(soa, svlen_thres, suppread_thres, n_max)
        blk = np.ascontiguousarray(blk, dtype=np.uint8)
        rb = (5 * int(n_max) + 15) // 16 * 16
        out = np.zeros(block_bytes(n_max, n_slots), dtype=np.uint8)
        out[:rb] = blk[:rb]
        out[rb:rb + 4] = np.array([status], dtype=np.uint32).view(np.uint8)
        if n_slots and soa.n_cands and status == 0:
            pred = blk[4 * int(n_max):4 * int(n_max) + soa.n_cands]
            kept = np.bincount(np.asarray(cand_slot, dtype=np.int64)[pred != 0], minlength=int(n_slots)).astype(np.uint64)
            out[rb + 16:] = kept[:int(n_slots)].view(np.uint8)
        return self.allgather(out)

    def close(self):
        pass


class RcclGather(object):
    """The in-library collective: ncclAllGather on device buffers of `ctx`'s device (duet_comm_* of include/duet_ef.h)."""
    name = 'rccl'

    def __init__(self, ctx, star):
        import ctypes
        self.ctx, self.star = ctx, star
        lib = ctx.lib
        # The hand-over is symmetric: rank 0 ALWAYS broadcasts -- the 128-byte id, or why it has none -- so that no rank is left
        # inside the broadcast while rank 0 has moved on to another collective (an RCCL that cannot be loaded on
        # rank 0 used to hang the others until the rendezvous timeout); every rank then raises the same error.
        ident = None
        if star.rank == 0:
            buf = (ctypes.c_ubyte * 128)()
            rc = lib.duet_comm_unique_id(ctx.handle, buf)
            ident = (b'\x00' + bytes(buf)) if rc == 0 else b'\x01' + ('duet_comm_unique_id: %s' % ctx.last_error()).encode('utf-8', 'replace')
        ident = star.bcast(ident)
        if not isinstance(ident, (bytes, bytearray)) or len(ident) != 129 or ident[:1] != b'\x00':
            why = bytes(ident)[1:].decode('utf-8', 'replace') if ident else 'rank 0 sent no id'
            raise CommError(why or 'rank 0 sent no id')
        arr = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(ident)[1:])
        # bounded inside the library by DUET_RDZV_TIMEOUT (ncclCommInitRank on a helper thread, waited for with a deadline)
        os.environ['DUET_RDZV_TIMEOUT'] = repr(float(star.timeout))
        self.handle = lib.duet_comm_create(ctx.handle, arr, star.rank, star.world)
        if not self.handle:
            why = ctx.last_error()
            raise (CommTimeout if 'did not finish within' in why else CommError)('duet_comm_create: %s' % why)
        if os.environ.get('DUET_COMM_SELFTEST') == '1':
            # before any result travels: a rank-stamped pattern through the same communicator, every slot checked on every rank
            self.selftest()

    def _raise(self, rc):
        from duet_amd import _lib
        if rc == _lib.DUET_ERR_TIMEOUT:
            raise CommTimeout(self.ctx.last_error())
        self.ctx._raise(rc)

    def ef_allgather(self, soa, svlen_thres, suppread_thres, cand_slot, n_slots, n_max, compute=None):
        """duet_comm_ef_allgather: upload, ef_classify -> ef_seed_sort -> ef_finalize into the rank's block on the device,
        trailer kernel, ONE ncclAllGather on the same stream, one download.  -> uint8 [world, block_bytes(n_max, n_slots)]
        (`compute` is not used: the kernels run inside the call)."""
        import ctypes
        from duet_amd import _lib
        prob, keep = _lib.problem_from_arrays(soa, svlen_thres, suppread_thres)
        slots = np.ascontiguousarray(cand_slot, dtype=np.uint32)
        out = np.empty((self.star.world, block_bytes(n_max, n_slots)), dtype=np.uint8)
        rc = self.ctx.lib.duet_comm_ef_allgather(self.handle, ctypes.byref(prob), slots.ctypes.data_as(ctypes.c_void_p) if slots.size else None,
                                                 ctypes.c_uint32(int(n_slots)), ctypes.c_uint32(int(n_max)),
                                                 out.ctypes.data_as(ctypes.c_void_p))
        del keep
        if rc:
            self._raise(rc)
        return out

    def allgather(self, block):
        import ctypes
        block = np.ascontiguousarray(block, dtype=np.uint8)
        out = np.empty((self.star.world, block.size), dtype=np.uint8)
        rc = self.ctx.lib.duet_comm_allgather_host(self.handle, block.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(block.size),
                                                   out.ctypes.data_as(ctypes.c_void_p))
        if rc:
            self._raise(rc)
        return out

    def allgather_device(self, send_ptr, nbytes, recv_ptr, stream):
        """device pointers, asynchronous on `stream` (duet_comm_allgather_device)"""
        import ctypes
        rc = self.ctx.lib.duet_comm_allgather_device(self.handle, ctypes.c_void_p(send_ptr), ctypes.c_uint64(int(nbytes)),
                                                     ctypes.c_void_p(recv_ptr), ctypes.c_void_p(stream))
        if rc:
            self._raise(rc)

    def info(self):
        """What RCCL itself reports about the communicator (duet_comm_info): {'rank', 'world', 'rccl_ranks', 'rccl_rank',
        'rccl_device'} -- rccl_ranks == world is the evidence that RCCL connected every rank."""
        import ctypes
        v = [ctypes.c_int(-1) for _ in range(5)]
        rc = self.ctx.lib.duet_comm_info(self.handle, *[ctypes.byref(x) for x in v])
        if rc:
            self._raise(rc)
        return dict(zip(('rank', 'world', 'rccl_ranks', 'rccl_rank', 'rccl_device'), (int(x.value) for x in v)))

    def selftest(self, words=4096):
        """Every rank gathers a rank-stamped pattern and checks every slot (duet_comm_selftest; collective)."""
        rc = self.ctx.lib.duet_comm_selftest(self.handle, int(words))
        if rc:
            self._raise(rc)

    def close(self):
        if self.handle:
            self.ctx.lib.duet_comm_destroy(self.handle)
            self.handle = None
