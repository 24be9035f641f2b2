"""The shared library loads without a GPU and exports every function include/marl_amd.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "marl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(marl_[a-z_]+)\s*\(", text)))


def test_header_and_library_agree():
    from marl_amd import _lib

    names = _declared()
    assert len(names) >= 15
    assert sorted(_lib.EXPORTS) == names, "marl_amd/_lib.py binds a different set than the header declares"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"libmarl_amd.so does not export {missing}"


def test_argument_errors_are_reported_without_touching_the_gpu():
    from marl_amd import _lib

    L = _lib.lib()
    handle = ctypes.c_void_p()
    rows = (ctypes.c_char_p * 3)(b"###", b"# #", b"###")
    # more than 4 agents are not supported by the lane mapping; rejected before any HIP call
    rc = L.marl_env_create(ctypes.byref(handle), rows, 3, 5, 0, 1000, 4, 0, 0, 0)
    assert rc == -1 and b"num_agents" in L.marl_last_error()
    rc = L.marl_env_create(ctypes.byref(handle), rows, 3, 2, 9, 1000, 4, 0, 0, 0)
    assert rc == -1 and b"feature" in L.marl_last_error()
    bad = (ctypes.c_char_p * 3)(b"###", b"#x#", b"###")
    rc = L.marl_env_create(ctypes.byref(handle), bad, 3, 1, 0, 1000, 4, 0, 0, 0)
    assert rc == -2 and b"unknown layout character" in L.marl_last_error()
    open_grid = (ctypes.c_char_p * 3)(b"# #", b"# #", b"###")
    rc = L.marl_env_create(ctypes.byref(handle), open_grid, 3, 1, 0, 1000, 4, 0, 0, 0)
    assert rc == -2 and b"enclosed" in L.marl_last_error()
    rc = L.marl_gae(None, None, None, None, 4, 4, 0.99, 0.95, None, None, 0, None, 0, None)
    assert L.marl_gae_workspace_bytes(128, 131072) == 128 * 4 + 128 * 16 * 8
    assert rc == -1
    assert L.marl_env_obs_dim(None) == -1


def _prototypes():
    """{name: (return kind, [parameter kinds])} parsed from the header; kinds: ptr, int, u64, i64, f64, size, void"""
    text = open(os.path.join(ROOT, "include", "marl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)

    def kind(decl):
        decl = decl.strip()
        if "*" in decl:
            return "ptr"
        for word, k in (("uint64_t", "u64"), ("int64_t", "i64"), ("double", "f64"), ("size_t", "size"), ("int", "int"),
                        ("void", "void")):
            if re.search(r"\b%s\b" % word, decl):
                return k
        raise AssertionError(f"cannot classify {decl!r}")

    out = {}
    for ret, name, params in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(marl_[a-z_]+)\s*\(([^)]*)\)\s*;", text):
        plist = [kind(p) for p in params.split(",")] if params.strip() not in ("", "void") else []
        out[name] = (kind(ret), plist)
    return out


def test_binding_matches_the_header_parameter_by_parameter():
    """The ctypes signatures in marl_amd/_lib.py against the prototypes of include/marl_amd.h: the same
    number of parameters, each of the same machine kind, the same kind of return value."""
    from marl_amd import _lib

    L = _lib.lib()
    protos = _prototypes()
    assert sorted(protos) == sorted(_lib.EXPORTS)

    def ckind(t):
        if t is None:
            return "void"
        if t in (ctypes.c_void_p, ctypes.c_char_p) or isinstance(t, type(ctypes.POINTER(ctypes.c_int))) and issubclass(t, ctypes._Pointer):
            return "ptr"
        return {ctypes.c_int: "int", ctypes.c_uint64: "u64", ctypes.c_int64: "i64", ctypes.c_double: "f64",
                ctypes.c_size_t: "size"}[t]

    for name, (ret, params) in protos.items():
        fn = getattr(L, name)
        assert fn.argtypes is not None, f"{name}: no argtypes set"
        got = [ckind(t) for t in fn.argtypes]
        # size_t and uint64_t are the same machine type here; ctypes may alias them
        norm = lambda ks: ["u64" if k == "size" else k for k in ks]  # noqa: E731
        assert norm(got) == norm(params), f"{name}: binding {got} vs header {params}"
        assert norm([ckind(fn.restype)]) == norm([ret]), f"{name}: return {ckind(fn.restype)} vs header {ret}"
