import os
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


_REAL_STDERR = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Make a process abort self-describing (VERDICT r3 #1c).  pytest redirects file descriptor 2 into a temporary file per
    # test; the ROCm runtime reports a GPU fault as a line on stderr + abort() from its event thread, and glibc's heap checks
    # likewise — lines that die with the process inside that file, leaving only faulthandler's "Aborted" at whatever
    # synchronising call the main thread was in.  While pytest_configure runs, capture is suspended and fd 2 is the real
    # stderr: duplicate it now; the library's abort trace (mrirt_install_abort_trace, installed once the library is loaded)
    # writes the aborting thread's native stack and the tail of the capture file there.
    global _REAL_STDERR
    os.environ.setdefault("LIBC_FATAL_STDERR_", "1")           # glibc: fatal messages to stderr, not /dev/tty
    os.environ.setdefault("HSA_ENABLE_DEBUG", "0")
    try:
        _REAL_STDERR = os.dup(2)
        os.set_inheritable(_REAL_STDERR, False)
    except OSError:
        _REAL_STDERR = None


def install_abort_trace():
    """Load the built library (no GPU call) and point its SIGABRT trace at the real stderr; returns whether it is armed."""
    if _REAL_STDERR is None:
        return False
    import mrirt
    if not mrirt._lib.SO_PATH.exists():
        return False
    return mrirt._lib.lib().mrirt_install_abort_trace(_REAL_STDERR) == 0


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _fresh_hip_library():
    """On a GPU box, make sure the in-tree libmrirt.so matches the sources before any test loads it (a no-op
    when it is up to date; the snapshot normally carries the library __graft_entry__.build() produced); on a box
    without a GPU, build it when it is missing altogether (a fresh clone: the ABI tests dlopen it).  The product
    itself never builds or falls back implicitly — this is test hygiene only."""
    try:
        import torch
        import mrirt
        if torch.cuda.is_available() or not mrirt._lib.SO_PATH.exists():
            mrirt._lib.build()
    except Exception as e:                      # no hipcc on the box: the tests will say what is missing
        print(f"[conftest] libmrirt.so not rebuilt: {e}")
    try:
        install_abort_trace()
    except Exception as e:
        print(f"[conftest] abort trace not installed: {e}")
    yield


# ---- fault attribution (opt-in: MRIRT_TRACE_CALLS=1) ---------------------------------------------------------
# With AMD_SERIALIZE_KERNEL=3 the runtime waits for every kernel, so a GPU fault surfaces inside the call that launched the
# faulting kernel.  This writes, to the (captured) stderr whose tail the abort trace prints: per test the caching allocator's
# segments, and per C-ABI call its name and integer / pointer arguments — enough to name the kernel and to place the fault
# address relative to the buffers it was given.
_TRACE_FH = None


def _trace_write(text: str) -> None:
    """To the captured stderr (whose tail the abort trace prints) and, when MRIRT_TRACE_FILE names one, to a file that
    survives a passing run too."""
    global _TRACE_FH
    os.write(2, text.encode())
    path = os.environ.get("MRIRT_TRACE_FILE")
    if path:
        if _TRACE_FH is None:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            _TRACE_FH = open(path, "a", buffering=1)
        _TRACE_FH.write(text)


class _TracedLib:
    def __init__(self, handle):
        object.__setattr__(self, "_h", handle)

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if not callable(fn) or not name.startswith("mrirt_"):
            return fn

        def call(*args):
            def show(a):
                v = getattr(a, "value", a)
                return hex(v) if isinstance(v, int) and v > 0xFFFF else str(v) if isinstance(v, (int, float, type(None))) else type(a).__name__
            _trace_write(f"[call] {name}(" + ", ".join(show(a) for a in args) + ")\n")
            return fn(*args)
        return call

    def __setattr__(self, name, value):
        setattr(self._h, name, value)


@pytest.fixture(autouse=True)
def _trace_calls(request):
    if not os.environ.get("MRIRT_TRACE_CALLS"):
        yield
        return
    import torch
    import mrirt
    real = mrirt._lib.lib()
    if not isinstance(mrirt._lib._LIB, _TracedLib):
        mrirt._lib._LIB = _TracedLib(real)
    mrirt.render.TRACE_HOOK = _trace_write
    lines = [f"[test] {request.node.nodeid}"]
    if torch.cuda.is_available():
        segs = sorted((s["address"], s["total_size"], sum(b["size"] for b in s["blocks"] if b["state"] != "inactive")) for s in torch.cuda.memory_snapshot())
        lines += [f"[seg] {a:#x}..{a + n:#x} ({n >> 20} MiB, {u >> 10} KiB in use)" for a, n, u in segs]
    _trace_write("\n".join(lines) + "\n")
    yield
