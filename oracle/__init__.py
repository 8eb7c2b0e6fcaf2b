"""CPU oracle for the denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain fp32 torch-CPU functional code, the arithmetic of the reference's
UNet denoiser, DDIM sampler and VAE decoder (each function cites the reference file:line it
follows).  It is pinned against golden vectors produced by importing the reference itself in the
authoring container (``oracle/make_golden.py`` -> ``tests/golden/``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.  The product package never imports
it and has no CPU fallback: without the HIP library the product raises.
"""
