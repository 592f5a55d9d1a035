"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the Myria3D RandLA-Net hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.  The product path
(``myria3d_amd``) never imports this package and fails loudly when its HIP library is missing.

PARITY UNPINNED BY THE REFERENCE: the reference implementation of this path lives in un-vendored
third-party wheels (torch_geometric 2.4, torch_cluster, torch_scatter — ``environment.yml:14-22``)
that are not installed here, and the reference's own tests pin only output *shapes*
(``tests/myria3d/models/modules/test_randla_nets.py:8-40``).  The oracle is therefore pinned by
(i) those shape cases, (ii) cross-checks of its kNN against ``scipy.spatial.cKDTree`` and of its
BatchNorm/Linear against stock ``torch.nn`` modules, and (iii) golden vectors it generated itself
(``tests/golden/``, generator script committed next to them).
"""
