"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the Myria3D RandLA-Net hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.  The product path
(``myria3d_amd``) never imports this package and fails loudly when its HIP library is missing.

PARITY: pinned to the reference's own file, third-party semantics restated.  The reference implementation of this path
lives in un-vendored third-party wheels (torch_geometric 2.4, torch_cluster, torch_scatter — ``environment.yml:14-22``)
that are not installed here, and the reference's own tests pin only output *shapes*
(``tests/myria3d/models/modules/test_randla_nets.py:8-40``).  Since round 3 the reference's
``myria3d/models/modules/pyg_randla_net.py`` itself RUNS in this container on ``tests/_pyg_stub`` (a restatement of the
six symbols it imports) and ``tests/test_reference_pin.py`` diffs its logits, loss, gradients,
running statistics and level-1 graph against this oracle (``tests/golden/randla_reference.npz`` carries them to the GPU
box).  What stays unpinned until the real wheels exist: the semantics of those six third-party calls.  Other pins:
(i) the reference's shape cases, (ii) cross-checks of the kNN against ``scipy.spatial.cKDTree`` and of BatchNorm/Linear
against stock ``torch.nn`` modules, (iii) golden vectors the oracle generated itself (``tests/golden/``).
"""
