"""Mirror of the reference's `network` package (operations, layers, upsampler, model_loss)."""
