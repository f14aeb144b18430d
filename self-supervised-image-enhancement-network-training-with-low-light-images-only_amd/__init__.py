"""MI355X-native hot path of SS-HSLIE (self-supervised hyperspectral low-light enhancement).

The directory name is not a Python identifier; import it through the repo-root `ssie.py`
loader (`import ssie; pkg = ssie.load()`), which registers it as module `ssie_amd`.
Everything here runs on hand-written HIP kernels in `lib/libssie_hip.so` (built from
`csrc/` by `build.py`); there is no CPU or PyTorch-operator fallback.
"""
__all__ = ["build", "hostlib"]
