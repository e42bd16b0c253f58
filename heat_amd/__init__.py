"""heat_amd — MI355X-native SimpleX/CCL collaborative-filtering training engine.

One hot path of visuOwO/HEAT (the per-interaction fused forward+backward+SGD step and its epoch loop),
rebuilt as hand-written gfx950 HIP kernels behind the reference's own `cf_c` module surface.

    heat_amd.abi      ctypes view of the C ABI (include/heat_cf.h)  -> lib/libheat_cf.so
    heat_amd.cf_c     pybind11 module with the reference's `cf_c.modules.*` classes (built from csrc/)
    heat_amd.cf       the reference's Python frontend mirrored (config, datasets, models, train, metrics, main)
    heat_amd.build    compiles csrc/ with hipcc for gfx950

There is no CPU fallback anywhere in this package: without the built HIP library, imports of the compute
entry points raise, and without a GPU every compute call returns HEAT_CF_EHIP.
"""
__all__ = ["abi", "build"]
