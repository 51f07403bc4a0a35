"""Context wrapper over the C ABI (include/sora_adsb_gpu.h)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BYTES_PER_SAMPLE, Config, Counts, EMIT_ACCEPTED, FRAME_DTYPE, WIRE_DTYPE,
                   check, lib)


def unpack_frames(wire, encoding):
    """28-byte wire frames (SORA_OPT_WIRE_FRAMES) -> FRAME_DTYPE, on the host."""
    wire = np.ascontiguousarray(wire)
    out = np.zeros(wire.size, dtype=FRAME_DTYPE)
    check(lib().sora_adsb_unpack_frames(wire.ctypes.data if wire.size else None, wire.size,
                                        int(encoding),
                                        out.ctypes.data if wire.size else None))
    return out


def _ptr(buf):
    """Device address of a DeviceBuffer, a torch CUDA tensor or a raw int."""
    if isinstance(buf, DeviceBuffer):
        return buf.ptr
    if isinstance(buf, int):
        return buf
    if hasattr(buf, "data_ptr"):        # torch tensor
        if not buf.is_cuda:
            raise ValueError("expected a device tensor")
        if not buf.is_contiguous():
            raise ValueError("expected a contiguous tensor")
        return buf.data_ptr()
    raise TypeError(f"cannot take a device pointer from {type(buf)!r}")


class DeviceBuffer:
    """hipMalloc'd bytes, for callers that do not want torch."""

    def __init__(self, nbytes, device=0):
        self.device = device
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().sora_adsb_device_alloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, data, device=0):
        arr = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8))
        buf = cls(arr.size, device)
        check(lib().sora_adsb_copy_to_device(device, buf.ptr, arr.ctypes.data, arr.size))
        return buf

    def to_host(self, nbytes=None):
        n = self.nbytes if nbytes is None else int(nbytes)
        out = np.empty(n, dtype=np.uint8)
        check(lib().sora_adsb_copy_to_host(self.device, out.ctypes.data, self.ptr, n))
        return out

    def free(self):
        if self.ptr:
            lib().sora_adsb_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class AdsbContext:
    """One device, one stream of samples (struct sora_adsb_ctx)."""

    def __init__(self, max_chunk_samples, device=0, emit_mode=EMIT_ACCEPTED,
                 max_candidates=0, max_squitters=0, max_frames=0, stream=None,
                 correct_single_bit=False, fixed_lists=False, wire_frames=False,
                 resolve_stream=None):
        self.device = device
        options = _lib.OPT_CORRECT_SINGLE_BIT if correct_single_bit else 0
        if fixed_lists:
            options |= _lib.OPT_FIXED_LISTS
        if wire_frames:
            options |= _lib.OPT_WIRE_FRAMES
        # what collect() and friends hand out: 48-byte frames, or 28-byte wire
        # frames (unpack_frames() turns them into the former)
        self.wire = bool(wire_frames)
        self.frame_dtype = WIRE_DTYPE if wire_frames else FRAME_DTYPE
        cfg = Config(int(max_chunk_samples), int(max_candidates), int(max_squitters),
                     int(max_frames), int(emit_mode), options, stream)
        h = C.c_void_p()
        rc = lib().sora_adsb_create(C.byref(h), device, C.byref(cfg))
        self._h = h.value
        if rc != _lib.OK:
            detail = lib().sora_adsb_last_error(self._h).decode() if self._h else ""
            if self._h:
                lib().sora_adsb_destroy(self._h)
                self._h = None
            raise _lib.SoraError(rc, detail)
        if resolve_stream is not None:
            self.set_resolve_stream(resolve_stream)

    def set_resolve_stream(self, stream):
        """Exchange, resolve and the copies of collect on `stream` (a raw
        hipStream_t; None: back on the context's own)."""
        check(lib().sora_adsb_set_resolve_stream(self._h, stream), self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().sora_adsb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:                    # at interpreter shutdown the module globals may be gone
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        check(lib().sora_adsb_reset(self._h), self._h)

    # -- stage 1 / exchange / stage 2 ------------------------------------
    def scan(self, d_iq, nbytes, encoding, base_position=0, n_owned=None):
        n_samples = nbytes // BYTES_PER_SAMPLE[encoding]
        if n_owned is None:
            n_owned = n_samples
        check(lib().sora_adsb_scan(self._h, _ptr(d_iq), nbytes, encoding, base_position,
                                   n_owned), self._h)

    def squitter_count(self):
        n = C.c_uint64()
        check(lib().sora_adsb_squitter_count(self._h, C.byref(n)), self._h)
        return n.value

    def copy_squitters(self, d_dst, cap_records):
        check(lib().sora_adsb_copy_squitters(self._h, _ptr(d_dst), cap_records), self._h)

    def add_known(self, d_records, n_records):
        check(lib().sora_adsb_add_known(self._h, _ptr(d_records), n_records), self._h)

    def export_block(self, d_block, block_records):
        """Exchange block (count, then records) into device memory; no host sync."""
        check(lib().sora_adsb_export_block(self._h, _ptr(d_block), block_records), self._h)

    def add_known_blocks(self, d_blocks, n_blocks, block_records, skip_block=-1):
        """Merges gathered exchange blocks; counts are read on the device."""
        check(lib().sora_adsb_add_known_blocks(self._h, _ptr(d_blocks), n_blocks,
                                               block_records, skip_block), self._h)

    def resolve(self):
        check(lib().sora_adsb_resolve(self._h), self._h)

    def submit_device(self, d_iq, nbytes, encoding, base_position=0, n_owned=None):
        n_samples = nbytes // BYTES_PER_SAMPLE[encoding]
        if n_owned is None:
            n_owned = n_samples
        check(lib().sora_adsb_submit_device(self._h, _ptr(d_iq), nbytes, encoding,
                                            base_position, n_owned), self._h)

    def submit_host(self, data, encoding, base_position=0, n_owned=None):
        arr = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8))
        n_samples = arr.size // BYTES_PER_SAMPLE[encoding]
        if n_owned is None:
            n_owned = n_samples
        check(lib().sora_adsb_submit_host(self._h, arr.ctypes.data, arr.size, encoding,
                                          base_position, n_owned), self._h)

    # -- results ----------------------------------------------------------
    def stream(self, source, encoding, block_samples):
        """Streaming ingest (sora_adsb_stream_*): `source` is a bytes-like object
        or a binary file object; yields one FRAME_DTYPE array per block."""
        failure = []        # ctypes swallows an exception raised inside a callback:
                            # keep it, end the stream, raise it afterwards

        if hasattr(source, "readinto"):
            def read_some(dst, nbytes):
                buf = (C.c_char * nbytes).from_address(dst)
                return source.readinto(buf) or 0
        else:
            view = memoryview(source).cast("B")
            at = [0]

            def read_some(dst, nbytes):
                k = min(nbytes, len(view) - at[0])
                if k > 0:
                    C.memmove(dst, (C.c_char * k).from_buffer_copy(view[at[0]:at[0] + k]), k)
                at[0] += k
                return k

        def read(_user, dst, nbytes):
            if failure:
                return 0
            try:
                return read_some(dst, nbytes)
            except BaseException as e:          # noqa: BLE001 -- re-raised below
                failure.append(e)
                return 0
        cb = _lib.READ_FN(read)
        check(lib().sora_adsb_stream_begin(self._h, int(encoding), int(block_samples),
                                           cb, None), self._h)
        try:
            while True:
                ptr, n = C.c_void_p(), C.c_size_t()
                rc = lib().sora_adsb_stream_next(self._h, C.byref(ptr), C.byref(n))
                if failure:
                    raise failure[0]    # the source failed: not the end of the stream
                if rc == _lib.END:
                    break
                check(rc, self._h)
                if n.value:
                    raw = C.string_at(ptr.value, n.value * FRAME_DTYPE.itemsize)
                    yield np.frombuffer(raw, dtype=FRAME_DTYPE).copy()
                else:
                    yield np.empty(0, dtype=FRAME_DTYPE)
        finally:
            lib().sora_adsb_stream_end(self._h)

    def _collect_fn(self, device=False):
        L = lib()
        if self.wire:
            return L.sora_adsb_collect_wire_device if device else L.sora_adsb_collect_wire
        return L.sora_adsb_collect_device if device else L.sora_adsb_collect

    def collect(self, cap=None):
        """Frames of the last resolve as a numpy structured array (self.frame_dtype)."""
        n = C.c_size_t()
        fn = self._collect_fn()
        if cap is None:
            # ask for the count first, then fetch exactly that many
            check(fn(self._h, None, 0, C.byref(n)), self._h)
            cap = n.value
        out = np.zeros(cap, dtype=self.frame_dtype)
        check(fn(self._h, out.ctypes.data if cap else None, cap, C.byref(n)), self._h)
        return out[:min(cap, n.value)]

    def collect_into(self, out):
        """Frames into a preallocated (ideally pinned) numpy array; returns count."""
        n = C.c_size_t()
        assert out.dtype == self.frame_dtype
        check(self._collect_fn()(self._h, out.ctypes.data, out.size, C.byref(n)), self._h)
        return n.value

    def frame_count(self):
        """Frames the last resolve produced (waits for it; lists that overflowed
        are grown and the chunk run again first)."""
        n = C.c_size_t()
        check(self._collect_fn()(self._h, None, 0, C.byref(n)), self._h)
        return n.value

    def collect_device(self, d_out, cap):
        n = C.c_size_t()
        check(self._collect_fn(device=True)(self._h, _ptr(d_out), cap, C.byref(n)), self._h)
        return n.value

    def counts(self):
        c = Counts()
        rc = lib().sora_adsb_get_counts(self._h, C.byref(c))
        if rc not in (_lib.OK, _lib.ERR_OVERFLOW):
            check(rc, self._h)
        d = {k: getattr(c, k) for k, _ in Counts._fields_}
        d["overflow"] = rc == _lib.ERR_OVERFLOW
        return d

    def list_regrows(self):
        """Chunks this context ran again with larger lists after an overflow."""
        return lib().sora_adsb_list_regrows(self._h)

    def synchronize(self):
        check(lib().sora_adsb_synchronize(self._h), self._h)

    def last_scan_ms(self):
        ms = C.c_float()
        check(lib().sora_adsb_last_scan_ms(self._h, C.byref(ms)), self._h)
        return ms.value

    def scan_kernel_name(self, encoding):
        return lib().sora_adsb_scan_kernel_name(self._h, encoding).decode()

    def magnitude(self, d_iq, nbytes, encoding, d_m):
        check(lib().sora_adsb_magnitude(self._h, _ptr(d_iq), nbytes, encoding, _ptr(d_m)),
              self._h)
