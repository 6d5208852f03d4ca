"""Test / bench scaffolding, NOT part of the product package (moved out of gigapose_amd/ in round 6): the model wired as the reference's
Hydra config wires it with deterministic random-init weights (`factory`; there is no network for gigaPose_v1.ckpt or the hub DINOv2),
and synthetic template sets, crops and geometry of the shapes BASELINE.json names (`synthetic`).  Importers: tests/, tools/, bench.py,
__graft_entry__.smoke(), oracle/ (generators of the golden vectors)."""
