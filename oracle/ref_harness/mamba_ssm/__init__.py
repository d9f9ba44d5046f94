"""Stand-in provider for the un-vendored third-party `mamba_ssm` package (pinned by the reference at
mamba-ssm==1.2.0.post1, caduceus_env.yml:50) so that the reference's own `caduceus/*.py` can be imported
IN THE BUILD CONTAINER ONLY to generate golden fixtures (oracle/gen_golden.py).

This is our own code (an adapter around the installed third-party `transformers` MambaMixer torch path plus
small torch restatements of upstream's Block / RMSNorm / rms_norm_fn semantics). It is test infrastructure:
nothing in caduceus_amd/ imports it and it never runs on the GPU box.
"""
__version__ = "1.2.0.post1+shim"
