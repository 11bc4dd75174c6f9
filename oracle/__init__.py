"""CPU oracle for the Diff-MST mix-console hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline - never as
the thing shipped.  The product path (``diff-mst_amd/diffmst_hip`` + the HIP library)
must fail loudly when the HIP extension is missing; it never falls back here.

PARITY STATUS
-------------
* mst-owned logic (parameter slicing / denormalisation / op order / bus sum /
  AudioFeatureLoss features / Bark filterbank / peak normalise):
  **pinned** - ``tests/golden/make_golden.py`` imports the real reference
  modules from ``/root/reference`` (through ``sys.modules`` stubs for the absent
  third-party packages) and asserts this restatement equals them; the vectors it
  writes are committed under ``tests/golden/``.
* third-party arithmetic (``dasp-pytorch==0.0.1`` functional ops,
  ``auraloss==0.4.0`` STFT loss): **parity unpinned**.  Neither package exists
  in this image, in the wheelhouse, or under ``/root/reference``; there are no
  golden vectors for them in the reference's tests (which hold no assertions).
  ``dasp_restated.py`` / ``loss_restated.py`` restate the published
  algorithms and are anchored on the reference's call
  sites (``mst/modules.py:231-312``, ``configs/models/naive.yaml:54-68``) and
  on analytic known-answer tests (``tests/test_oracle_kat.py``).  The same holds for the fx bus added in
  round 2: ``noise_shaped_reverberation`` / ``octave_band_filterbank`` are restated from dasp-pytorch 0.0.1
   with the noise as an explicit input; the call site (``mst/modules.py:275-284``) is pinned by
  ``tests/golden/make_golden.py fx`` running the real ``AdvancedMixConsole`` over the restated dasp ops.
"""
