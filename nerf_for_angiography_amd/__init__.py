"""MI355X-native hot path of NeRF-for-angiography: fused ray generation -> sampling -> CPPN MLP ->
Beer-Lambert compositing (forward + backward) behind the reference's nerf/ and model/ call surface."""
__version__ = "0.1.0"
