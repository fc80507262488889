"""avatarcap_amd: MI355X-native (gfx950) hot path of AvatarCap's per-frame volumetric reconstruction.

Only what SURVEY.md section 8 scopes: the fused occupancy/colour MLP queries, marching cubes +
normals, KNN/LBS, their host-side mirror of the reference's Python interface, and the C-ABI
(`include/avcap.h`) they sit on.  See DESIGN.md.
"""
__version__ = '0.1.0'
