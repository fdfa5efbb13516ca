from .patch_embed import PatchEmbed

__all__ = ["PatchEmbed"]
