"""CPU oracle for the batched ATACOM env-step path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``rl_on_manifold_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may (as the checker / the reported CPU baseline, never as
the thing shipped).  It is a float64 numpy restatement of the reference's
algorithm for the hot path; every function
cites the reference ``file:line`` it follows.

Parity status
-------------
* pinned: A1-A8, A11-A13, A15(circle) -- checked against golden vectors
  captured from the reference's own Python modules imported in the build
  container (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
* parity unpinned: the Pinocchio / PyBullet arithmetic underneath A9, A10,
  A14, A15(planar, iiwa), A16(CLIK) -- those packages are not vendored in the
  reference and are absent from this image; the oracle restates their
  published algorithms and pins them with URDF known answers and finite
  differences only (see DESIGN.md section "Oracle").
"""
