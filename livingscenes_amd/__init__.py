"""livingscenes_amd -- MI355X-native (gfx950) implementation of the LivingScenes per-instance inference hot path
(VN-DGCNN+attention encoder, SDF decoder queries, instance matcher, Kabsch/ICP registration) behind the reference's
own Python call surface.  Device work = hand-written HIP behind a C ABI (include/livingscenes_hip.h); PyTorch is only
the allocator / stream provider.  There is no CPU fallback."""
__version__ = "0.1.0"
