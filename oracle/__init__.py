"""CPU oracle for the resample -> coadd -> subtract hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker.  The product path
(``zuds-pipeline_amd``) never imports it and fails loudly without the HIP
library.

PARITY UNPINNED.  The arithmetic of this path does not live in the reference
tree: ``zuds`` shells out to SWarp >= 2.38.0 (``zuds/constants.py:86-90``),
hotpants >= 5.1.11 (``zuds/constants.py:91-95``; un-vendored, empty submodule
``.gitmodules:1-3``) and SExtractor >= 2.18 (``zuds/constants.py:81-85``).
None of those binaries, nor astropy, exist in the build container or on the GPU
box, and the reference's only two known-answer stamps
(``zuds/tests/suite/test_stack.py:9-28``, ``zuds/tests/suite/test_sub.py:8-36``)
need network inputs.  What is restated here is therefore (a) the operator
*definition* the reference fixes by its flags and configs (cited per function)
and (b) the published algorithms of the three tools, each adopted as an explicit
convention and pinned by analytic known-answer tests in ``tests/``.  Pieces that
an independent implementation on this image can check are checked against it
(``tests/test_independent_pins.py``): the normalised Lanczos-3 kernel against
Pillow's resampler, bilinear interpolation against scipy.ndimage, the natural
bicubic spline of the background map against scipy.interpolate; WCS and FITS
against astropy-made vectors (``tests/golden``).  The tools' own conventions
(edges, snapping, masks, the mode estimator, all of hotpants) stay unpinned.

Everything is numpy float64 unless a function says otherwise.
"""
