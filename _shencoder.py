"""`_shencoder`: the second native module the reference's canonical MLP file imports.

/root/reference/core/nets/occnerf/canonical_mlps/occnerf_mlp.py:6 imports core.nets.occnerf.shencoder, whose
sphere_harmonics.py:9-12 does ``import _shencoder as _backend`` (src/bindings.cpp:5-8).  The encoder is never evaluated on the
rendering path (its only use, occnerf_mlp.py:46, is commented out in the reference; SURVEY.md section 2: out of scope), so the
module needs to IMPORT and nothing else: both entry points exist under the reference's names and signatures
(shencoder.h:9-10) and refuse by name when called.
"""


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx=None):
    raise NotImplementedError('_shencoder.sh_encode_forward: the spherical-harmonics encoder is outside the rendering path '
                              '(occnerf_mlp.py:46 is commented out in the reference) and is not built for gfx950')


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    raise NotImplementedError('_shencoder.sh_encode_backward: the spherical-harmonics encoder is outside the rendering path '
                              '(occnerf_mlp.py:46 is commented out in the reference) and is not built for gfx950')


__all__ = ['sh_encode_forward', 'sh_encode_backward']
