"""Level-A batch engine binding (include/zeekstd_amd.h): N frames in, N frames out on one MI355X."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib


class ZkError(Exception):
    def __init__(self, code, detail=""):
        self.code = code
        msg = _lib.error_name(code)
        super().__init__(f"{msg} (code {code}){': ' + detail if detail else ''}")


def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


class Engine:
    """Owns one GPU's scratch + stream. One thread at a time (like a libzstd context)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        rc = lib.zk_engine_create(device, C.byref(h))
        if rc != 0:
            raise ZkError(rc)
        self._h = h

    def close(self):
        if self._h:
            lib.zk_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_name(self) -> str:
        return lib.zk_engine_device_name(self._h).decode()

    def set_profiling(self, on: bool):
        rc = lib.zk_engine_set_profiling(self._h, int(on))
        if rc != 0:
            self._raise(rc)

    def set_fse_kernel(self, mode: int):
        """0 = by batch size, 1 = one lane per block, 2 = a quad of lanes per block (own-table blocks; same results)."""
        rc = lib.zk_engine_set_fse_kernel(self._h, int(mode))
        if rc != 0:
            self._raise(rc)

    CHOICES = {"reset": 0, "fse_own": 1, "fse_shared": 2, "exec_lanes": 3, "exec_ring": 4, "xxh64": 5, "small_path": 6,
               "pipe_contexts": 7, "pipe_chunk_mib": 8, "exec_resident": 9, "exec_seg": 10, "seg_kib": 11, "seg_fill": 12}

    def frame_content_sizes(self, comp: bytes, c_off, first=0, count=None):
        """zk_frame_content_sizes: the decompressed sizes of frames nobody holds seek entries for (header walk + sequence walks on the device,
        no output) -> (uint64 array, int32 status array: 0 or -ZSTD_ErrorCode per frame)"""
        c = _u64(c_off)
        count = len(c) - 1 - first if count is None else count
        src = np.frombuffer(bytes(comp) + b"\0" * 8, np.uint8)
        sizes = np.zeros(max(count, 1), np.uint64)
        st = np.zeros(max(count, 1), np.int32)
        rc = lib.zk_frame_content_sizes(self._h, src.ctypes.data, len(comp), c.ctypes.data, first, count, sizes.ctypes.data, st.ctypes.data)
        if rc != 0:
            self._raise(rc)
        return sizes[:count], st[:count]

    def checksums_followed(self):
        """Frames of the last finished decode that zk_k_xxh64_follow verified beside the executor (zk_engine_checksums_followed)."""
        return int(lib.zk_engine_checksums_followed(self._h))

    def set_kernel_choice(self, **kw):
        """Pins kernel variants (zk_engine_set_kernel_choice): e.g. set_kernel_choice(fse_shared=2, exec_lanes=256, xxh64=2) runs the
        large-batch kernels on whatever is decoded next; set_kernel_choice(reset=0) returns to "by batch shape"."""
        for key, value in kw.items():
            rc = lib.zk_engine_set_kernel_choice(self._h, self.CHOICES[key], int(value))
            if rc != 0:
                self._raise(rc)

    def kernel_times(self):
        """{kernel name: ms} of the last decode/encode call (profiling must be on)."""
        n = lib.zk_engine_kernel_count()
        ms = (C.c_float * n)()
        lib.zk_engine_kernel_times(self._h, ms, n)
        return {lib.zk_engine_kernel_name(k).decode(): float(ms[k]) for k in range(n) if ms[k] > 0}

    def _raise(self, rc):
        raise ZkError(rc, lib.zk_engine_last_hip_error(self._h).decode() if rc == -2001 else "")

    # ---- host-pointer entry points
    def decode_frames(self, comp, c_off, d_off, first=0, count=None, verify=True, raise_on_error=True, prefix=None):
        """Returns (bytes of frames [first, first+count), per-frame status array).
        prefix: raw-content prefix every frame was compressed against (patch mode)."""
        comp = np.frombuffer(comp, dtype=np.uint8) if not isinstance(comp, np.ndarray) else comp
        c_off, d_off = _u64(c_off), _u64(d_off)
        if count is None:
            count = len(c_off) - 1 - first
        out_len = int(d_off[first + count] - d_off[first])
        out = np.empty(max(out_len, 1), dtype=np.uint8)
        status = np.zeros(max(count, 1), dtype=np.int32)
        if prefix:
            pre = np.frombuffer(bytes(prefix), dtype=np.uint8)
            rc = lib.zk_decode_frames_prefix(self._h, comp.ctypes.data, comp.size, c_off.ctypes.data, d_off.ctypes.data,
                                             first, count, pre.ctypes.data, pre.size, out.ctypes.data, out_len, int(verify),
                                             status.ctypes.data)
        else:
            rc = lib.zk_decode_frames(self._h, comp.ctypes.data, comp.size, c_off.ctypes.data, d_off.ctypes.data,
                                      first, count, out.ctypes.data, out_len, int(verify), status.ctypes.data)
        if rc != 0 and (raise_on_error or rc <= -1000):
            self._raise(rc)
        return out[:out_len].tobytes(), status[:count]

    def encode_frames(self, data, frame_size=0x200000, level=1, checksum=False, prefix=None):
        """Returns (compressed payload bytes, [(c_size, d_size), ...]) -- one zstd frame per frame_size bytes.
        prefix: raw-content prefix referenced at the start of every frame (patch mode)."""
        data = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = int(data.size)
        nf = max(1, -(-n // frame_size))
        cap = int(lib.zk_compress_bound(n, frame_size))
        out = np.empty(cap, dtype=np.uint8)
        cs = np.zeros(nf, np.uint32)
        ds = np.zeros(nf, np.uint32)
        nfo = C.c_uint32()
        wr = C.c_uint64()
        if prefix:
            pre = np.frombuffer(bytes(prefix), dtype=np.uint8)
            rc = lib.zk_encode_frames_prefix(self._h, data.ctypes.data if n else None, n, frame_size, level, int(checksum),
                                             pre.ctypes.data, pre.size, out.ctypes.data, cap, cs.ctypes.data, ds.ctypes.data, nf,
                                             C.byref(nfo), C.byref(wr))
        else:
            rc = lib.zk_encode_frames(self._h, data.ctypes.data if n else None, n, frame_size, level, int(checksum),
                                      out.ctypes.data, cap, cs.ctypes.data, ds.ctypes.data, nf, C.byref(nfo), C.byref(wr))
        if rc != 0:
            self._raise(rc)
        return out[:wr.value].tobytes(), list(zip(cs[:nfo.value].tolist(), ds[:nfo.value].tolist()))

    def encode_frames_dev(self, d_src, n, frame_size, level, checksum, d_dst, dst_cap, d_c_sizes=None, d_d_sizes=None, stream=None):
        nfo = C.c_uint32()
        wr = C.c_uint64()
        rc = lib.zk_encode_frames_dev(self._h, self._ptr(d_src), n, frame_size, level, int(checksum), self._ptr(d_dst), dst_cap,
                                      self._ptr(d_c_sizes) if d_c_sizes is not None else None,
                                      self._ptr(d_d_sizes) if d_d_sizes is not None else None, C.byref(nfo), C.byref(wr), stream)
        if rc != 0:
            self._raise(rc)
        return nfo.value, wr.value

    def xxh64_frames(self, data, off):
        data = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        off = _u64(off)
        n = len(off) - 1
        out = np.zeros(max(n, 1), dtype=np.uint64)
        rc = lib.zk_xxh64_frames(self._h, data.ctypes.data if data.size else None, off.ctypes.data, n, out.ctypes.data)
        if rc != 0:
            self._raise(rc)
        return out[:n]

    # ---- device-pointer entry points (torch tensors or raw ints)
    @staticmethod
    def _ptr(t):
        return t.data_ptr() if hasattr(t, "data_ptr") else int(t)

    def decode_frames_dev(self, d_comp, comp_size, d_c_off, d_d_off, first, count, d_dst, dst_cap, verify=True,
                          d_status=None, stream=None):
        rc = lib.zk_decode_frames_dev(self._h, self._ptr(d_comp), comp_size, self._ptr(d_c_off), self._ptr(d_d_off),
                                      first, count, self._ptr(d_dst), dst_cap, int(verify),
                                      self._ptr(d_status) if d_status is not None else None, stream)
        if rc <= -1000:
            self._raise(rc)
        return rc

    def decode_frames_prefix_dev(self, d_comp, comp_size, d_c_off, d_d_off, first, count, d_prefix, prefix_len, d_dst, dst_cap,
                                 verify=True, d_status=None, stream=None):
        rc = lib.zk_decode_frames_prefix_dev(self._h, self._ptr(d_comp), comp_size, self._ptr(d_c_off), self._ptr(d_d_off),
                                             first, count, self._ptr(d_prefix) if prefix_len else None, prefix_len,
                                             self._ptr(d_dst), dst_cap, int(verify),
                                             self._ptr(d_status) if d_status is not None else None, stream)
        if rc <= -1000:
            self._raise(rc)
        return rc

    def decode_submit_dev(self, d_comp, comp_size, d_c_off, d_d_off, first, count, d_dst, dst_cap, verify=True, d_status=None):
        """Enqueue a batch on one of the engine's two decode contexts; returns the slot to pass to decode_wait()."""
        slot = C.c_int(-1)
        rc = lib.zk_decode_submit_dev(self._h, self._ptr(d_comp), comp_size, self._ptr(d_c_off), self._ptr(d_d_off),
                                      first, count, self._ptr(d_dst), dst_cap, int(verify),
                                      self._ptr(d_status) if d_status is not None else None, C.byref(slot))
        if rc != 0:
            if rc <= -1000:
                self._raise(rc)
            raise RuntimeError(f"zk_decode_submit_dev: {rc}")
        return slot.value

    def decode_wait(self, slot):
        rc = lib.zk_decode_wait(self._h, int(slot))
        if rc <= -1000:
            self._raise(rc)
        return rc

    def decode_frame_list_dev(self, d_comp, comp_size, d_c_off, d_d_off, d_ids, d_out_off, count, d_dst, dst_cap, verify=True,
                              d_status=None, stream=None):
        """Random-access batch: archive frames d_ids[i] -> d_dst + d_out_off[i] (all device-resident)."""
        rc = lib.zk_decode_frame_list_dev(self._h, self._ptr(d_comp), comp_size, self._ptr(d_c_off), self._ptr(d_d_off),
                                          self._ptr(d_ids), self._ptr(d_out_off), count, self._ptr(d_dst), dst_cap, int(verify),
                                          self._ptr(d_status) if d_status is not None else None, stream)
        if rc <= -1000:
            self._raise(rc)
        return rc

    def xxh64_frames_dev(self, d_data, d_off, count, d_out, stream=None):
        rc = lib.zk_xxh64_frames_dev(self._h, self._ptr(d_data), self._ptr(d_off), count, self._ptr(d_out), stream)
        if rc != 0:
            self._raise(rc)
