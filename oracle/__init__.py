"""CPU oracle: a plain torch-fp32 restatement of the reference's lip->speech inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under lip2speech_unit_amd/ imports this package; only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() do, and only as the checker.  Every function cites the reference
file:line it follows and works on a state_dict with the reference's key names.

Pinning (see DESIGN.md "Oracle"): the reference holds no golden vectors for this path.  The
functions here are pinned against outputs of the reference's own importable modules run in the build container
(tools/make_golden.py -> tests/golden/*.npz: ResEncoder, the ESPnet conformer Encoder, MelCodeGenerator).  The fairseq
TransformerEncoder and the beam-search loop live in an un-vendored dependency (fairseq @ afc77bd) and are "parity
unpinned" against the reference itself; they are cross-checked against an independent port (HuggingFace
HubertEncoderStableLayerNorm) and by property tests respectively.
"""
