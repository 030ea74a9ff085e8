"""MI355X-native implementation of the PathTracer rendering hot path (host-side Python mirror)."""
