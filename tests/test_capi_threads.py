"""Host-side thread rules of the C ABI that can be checked without a GPU: marl_last_error is
per thread, and argument errors raised concurrently on several threads do not mix."""
import ctypes
import threading


def test_last_error_is_per_thread():
    from marl_amd import _lib

    L = _lib.lib()
    rows = (ctypes.c_char_p * 3)(b"###", b"# #", b"###")
    seen = {}
    barrier = threading.Barrier(2)

    def worker(name, agents, feature, expect):
        h = ctypes.c_void_p()
        for _ in range(200):
            rc = L.marl_env_create(ctypes.byref(h), rows, 3, agents, feature, 1000, 4, 0, 0, 0)
            msg = L.marl_last_error()
            if rc != -1 or expect not in msg:
                seen[name] = (rc, msg)
                return
        barrier.wait()
        seen[name] = "ok"

    t1 = threading.Thread(target=worker, args=("a", 9, 0, b"num_agents"))
    t2 = threading.Thread(target=worker, args=("b", 2, 77, b"feature"))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert seen == {"a": "ok", "b": "ok"}
