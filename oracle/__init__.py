"""TEST INFRASTRUCTURE ONLY.

CPU (fp32, torch-CPU / numpy) restatements of the reference's hot-path algorithms, used as the
parity checker by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  Nothing under ``rsvld_amd`` (the product) imports this package.

Pinning: the reference ships no tests or golden vectors for this path, so every
oracle function here is pinned against outputs of the reference ITSELF, generated in the authoring
container by ``tests/golden/gen_*.py`` (which import /root/reference) and committed under
``tests/golden/*.npz``; ``tests/test_oracle_*.py`` re-checks the oracle against those fixtures.
"""
