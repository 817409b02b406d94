"""oracle/ -- CPU restatement of the MTG/DeepConvSep separation path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker /
the reported CPU baseline -- never as the thing that is measured or shipped.
``deepconvsep_amd`` (the product) never imports ``oracle``.

Contents
--------
``stft_np``    NumPy restatement of ``stft_norm`` / ``istft_norm`` /
               ``compute_file`` / ``compute_inverse``
               (reference ``transform.py:224-396``, script copy
               ``examples/dsd100/separate_dsd.py:24-111``).
``tiling_np``  NumPy restatement of both tilers and of the cross-fade
               overlap-add (``examples/dsd100/separate_dsd.py:114-169``,
               ``util.py:220-327``).
``net_ref``    torch-CPU float64 restatement of every ``build_ca`` graph on the
               separation path + the soft-mask expression
               (``examples/*/separate_*.py``), with the Lasagne / Theano
               semantics written out in ``net_ref``'s docstring.
``pipeline``   restatement of the ``train_auto`` separation block
               (``examples/dsd100/separate_dsd.py:275-311`` and siblings).
``score_np``   NumPy restatement of the score-informed front-end: ``expandMidi`` /
               ``getMidiNum`` / ``str2midi`` / ``slicefft_slices`` /
               ``remove_overlap`` (``util.py:126-191, 424-606``) and ``filterSpec``
               + the network-input products
               (``examples/bach10_scoreinformed/separate_bach10.py:172-200,
               500-527``).
``ref_exec``   executes the reference's OWN pure-NumPy function bodies by line
               range from ``/root/reference`` (build container only -- that
               tree does not exist on the GPU box).  Used to pin ``stft_np`` /
               ``tiling_np`` / ``score_np`` and to generate ``tests/golden/*.npz``.

Parity status
-------------
* STFT / iSTFT / tiling / overlap-add: PINNED.  ``tests/golden/*.npz`` are
  outputs of the reference's own code (``ref_exec``) on seeded inputs, and
  ``stft_np`` / ``tiling_np`` reproduce them bit-for-bit
  (``tests/test_oracle_golden.py``).
* Score-informed front-end: PINNED.  ``score_np`` reproduces bit-for-bit the note
  tables and masks of the reference's own ``expandMidi`` / ``getMidiNum`` /
  ``filterSpec`` (executed by ``ref_exec.score()`` with Python-2 shims;
  ``tests/golden/score_*.npz``).
* Network (``build_ca`` + mask): PARITY UNPINNED.  The arithmetic lives in
  Theano==0.9.0 and Lasagne (git master, unpinned) -- ``requirements.txt:1-2``
  of the reference -- neither of which is vendored, installed or installable
  here, and the reference ships no tests, golden vectors or weights.
  ``net_ref`` restates the published semantics of the Lasagne layers named at
  the reference's call sites and is cross-checked two independent ways
  (autograd VJP vs explicit transposed convolution / un-pooling).
"""
