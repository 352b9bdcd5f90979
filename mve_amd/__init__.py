"""mve_amd: MI355X-native depth-map reconstruction (MVE libs/dmrecon hot path).

The product is the HIP library behind ``include/mi_dmrecon.h`` (built in-tree as
``mve_amd/csrc/libmi_dmrecon.so``).  This Python package is the thin harness
around it: ctypes bindings (:mod:`mve_amd.api`), MVE scene-directory I/O
(:mod:`mve_amd.scene_io`) and the synthetic scene generator (:mod:`mve_amd.synth`).
"""
__version__ = "0.1.0"
