"""The seeded scenes behind tests/golden/*.npz (inputs are rebuilt, only outputs are stored)."""
import helpers
from renderer_amd import scenes


def _tex(size, seed):
    return dict(texels=scenes.checker_texture(size, seed), gen_mips=True, filtering=1, tiled=False, fmt=43)


SCENES = {
    # the reference's simplest scene (ten-instance ring, test_ktx shader), reduced
    "ring_ktx_160x90": lambda: scenes.config0(160, 90, shader=scenes.SHADER_TEST_KTX, tex_size=64, segments=24, rings=12),
    "ring_push_constants_160x90": lambda: scenes.config0(160, 90, shader=scenes.SHADER_PUSH_CONSTANTS, tex_size=16, segments=24, rings=12),
    # third sample's state: dummy shader, 4 samples per pixel
    "ring_dummy_msaa4_96x54": lambda: scenes.config0(96, 54, shader=scenes.SHADER_DUMMY, samples=4, tex_size=64, segments=16, rings=8),
    # equal-depth ties decided by submission order
    "soup_ties_96x64": lambda: helpers.soup_scene(96, 64, 500, 1234, depth_levels=3, n_draws=4),
    # near/far/guard-band clipping through the reference camera
    "soup_clipped_128x80": lambda: helpers.perspective_soup(128, 80, 300, 4321, near_cross=True, shader=scenes.SHADER_TEST_KTX,
                                                           texture=_tex(64, 9), pipeline_kw=dict(cull_mode=0)),
    # micro-triangle grid, the shape of the 1080p benchmark scene
    "grid_120x68": lambda: scenes.config1(120, 68, quads=90, tex_size=64),
}
