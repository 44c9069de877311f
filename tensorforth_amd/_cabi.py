"""ctypes binding generated from a C header (include/t4k.h or oracle/t4_oracle.h).

Every pointer parameter is passed as a raw address (c_void_p): callers hand over
``tensor.data_ptr()`` (device) or ``ndarray.ctypes.data`` (host).  No torch types cross
the boundary.
"""
import ctypes
import re

_SCALARS = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32,
    "unsigned": ctypes.c_uint, "void": None,
}
_HANDLES = {"t4k_stream_t", "t4k_event_t", "t4k_graph_t"}


def _ctype(decl):
    decl = decl.strip()
    if decl == "void" or decl == "":
        return None
    if "*" in decl:
        if decl.replace(" ", "").startswith("constchar*"):
            return ctypes.c_char_p
        return ctypes.c_void_p
    if decl.replace("const", "").split()[:3] == ["unsigned", "long", "long"]:
        return ctypes.c_ulonglong
    toks = [t for t in re.split(r"\s+", decl) if t not in ("const",)]
    # drop the parameter name if present
    base = toks[0]
    if base in _HANDLES:
        return ctypes.c_void_p
    if base in _SCALARS:
        return _SCALARS[base]
    raise ValueError("unknown C type in header: %r" % decl)


def parse_header(path, prefix):
    """Return {name: (restype, [argtypes])} for every `prefix*` function declared in `path`."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)        # preprocessor lines
    text = text.replace('extern "C" {', " ")
    out = {}
    pat = re.compile(r"([A-Za-z_][\w\s\*]*?)\b(%s\w+)\s*\(([^;{]*?)\)\s*;" % re.escape(prefix), re.S)
    for m in pat.finditer(text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # strip trailing parameter name for scalars ("int M" -> "int")
                if "*" not in a:
                    parts = a.split()
                    if len(parts) > 1 and parts[-1] not in _SCALARS and parts[-1] not in _HANDLES:
                        a = " ".join(parts[:-1])
                argtypes.append(_ctype(a))
        out[name] = (_ctype(ret), argtypes)
    return out


def bind(lib, decls):
    missing = []
    for name, (restype, argtypes) in decls.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = restype
        fn.argtypes = argtypes
    return missing
