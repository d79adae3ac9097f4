from .ps_vae_model import PSVAEModel  # noqa: F401
