"""Device context: one process drives one GPU (one rank per device)."""
import ctypes as C
import weakref

from . import _ffi


class Context:
    """Owns the HIP stream all kernels of its Searcher/Model handles run on.
    Replaces `tch::Device::cuda_if_available()` (model.rs:117): there is no CPU device here —
    construction fails when no GPU is visible."""

    def __init__(self, device_index=0):
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().pcv_init(int(device_index), C.byref(self._h)))
        self.device_index = int(device_index)
        # handles living on this context's device and stream (Searcher, Model, NativeComm): they have to
        # be destroyed before the context, whatever order the caller (or interpreter shutdown) picks
        self._children = weakref.WeakSet()

    def _register(self, child):
        self._children.add(child)

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("context already shut down")
        return self._h

    def synchronize(self):
        _ffi.check(_ffi.lib().pcv_synchronize(self.handle))

    @property
    def stream(self):
        return _ffi.lib().pcv_stream(self.handle)

    def set_stream(self, hip_stream, adopt=True):
        """adopt=True: queue all further work on the caller's hipStream_t (int; 0 = the device's default
        stream), e.g. `torch.cuda.current_stream().cuda_stream`, so that torch collectives order against
        it.  adopt=False: back to the context's own stream."""
        _ffi.check(_ffi.lib().pcv_set_stream(self.handle, C.c_void_p(hip_stream or None), 1 if adopt else 0))

    def alloc(self, n_bytes):
        """Device buffer (returns the device pointer as int)."""
        p = C.c_void_p()
        _ffi.check(_ffi.lib().pcv_device_alloc(self.handle, int(n_bytes), C.byref(p)))
        return p.value

    def free(self, dptr):
        _ffi.check(_ffi.lib().pcv_device_free(self.handle, C.c_void_p(dptr)))

    def to_host(self, dptr, n_bytes):
        import numpy as np

        out = np.empty(n_bytes, dtype=np.uint8)
        _ffi.check(_ffi.lib().pcv_copy_to_host(self.handle, out.ctypes.data, C.c_void_p(dptr), n_bytes))
        return out

    def to_device(self, dptr, array):
        import numpy as np

        a = np.ascontiguousarray(array)
        _ffi.check(_ffi.lib().pcv_copy_to_device(self.handle, C.c_void_p(dptr), a.ctypes.data, a.nbytes))

    def close(self):
        if self._h:
            for child in list(self._children):
                try:
                    child.close()
                except Exception:
                    pass
            _ffi.lib().pcv_shutdown(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def device_count():
    return _ffi.lib().pcv_device_count()
