"""Host-side helpers around the hot path: point-cloud I/O and checkpoints."""
