"""TEST INFRASTRUCTURE: bind mp_hsir_amd's ops to the CPU-emulated build of the kernel sources."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))


def bind_emulator():
    import build_emu
    import mp_hsir_amd._lib as L
    path = build_emu.build()
    L.use_library_for_tests(path)
    assert L.is_emulated()
    return L
