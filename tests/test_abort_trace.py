"""An abort inside a captured test must describe itself (VERDICT r3 #1c): the one unexplained SIGABRT of round 3 left only
faulthandler's "Aborted" in the log because whatever the runtime wrote to stderr went into pytest's per-test capture file and
died with the process.  A child pytest run writes a line to fd 2 and aborts from a non-Python thread; the parent checks that
the log carries the native stack, the captured line and the Python stack."""
import pathlib
import subprocess
import sys
import textwrap

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_an_abort_under_capture_shows_its_stderr_and_native_stack(tmp_path):
    case = tmp_path / "test_child_abort.py"
    case.write_text(textwrap.dedent('''
        import ctypes, os, threading
        def test_abort_from_a_runtime_thread():
            os.write(2, b"Memory access fault by GPU node-1 (pretend): the line a runtime prints before abort()\\n")
            libc = ctypes.CDLL(None)
            t = threading.Thread(target=libc.abort)      # not the main thread: like the HSA event thread
            t.start(); t.join()
    '''))
    conftest = tmp_path / "conftest.py"
    conftest.write_text(f"import sys; sys.path.insert(0, {str(ROOT)!r})\nfrom tests.conftest import pytest_configure, _fresh_hip_library  # noqa: F401\n")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", str(case)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "mrirt abort trace" in out, out[-3000:]
    assert "Memory access fault by GPU node-1 (pretend)" in out, out[-3000:]       # the captured stderr came through
    assert "abort" in out and "Fatal Python error: Aborted" in out, out[-3000:]      # native stack + faulthandler's Python stacks
