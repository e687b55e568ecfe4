"""TEST INFRASTRUCTURE ONLY: float64 CPU restatements of the hot path (physics.py, pyfly_restated.py, gym_restated.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(fixed-wing-gym_amd/gym_fixed_wing) never does and fails loudly without libfwgym.so.  Pinning status: gym_restated.py is
pinned against the verbatim reference module and the golden vectors under tests/golden/; physics.py is PARITY UNPINNED
against PyFly 0.1.2 (the dependency is absent and not installable here), see DESIGN.md section 2."""
