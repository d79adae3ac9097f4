"""MI355X-native (gfx950) implementation of PS-VAE's training hot path behind the
reference's own module API: data.bpe_dataset (collation), modules.{common_nn, encoder,
vae_piece_decoder, predictor}, pl_models.PSVAEModel.  The arithmetic runs in hand-written
HIP kernels reached through the C-ABI library csrc/libpsvae_hip.so (see include/psvae_hip.h);
there is no CPU fallback: using an op without the library or without a GPU raises."""

__all__ = ['data', 'modules', 'pl_models', 'utils', 'ops', 'hip']
