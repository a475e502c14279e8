from setuptools import setup

setup(
    name='patchgan-amd',
    version='0.2.2+mi355x.1',
    description='MI355X-native (gfx950) implementation of the patchGAN G+D training hot path',
    packages=['patchgan_amd', 'patchgan'],
    package_data={'patchgan_amd': ['libpatchgan_hip.so', 'csrc/*']},
    entry_points={'console_scripts': ['patchgan_train = patchgan_amd.train:patchgan_train',
                                      'patchgan_infer = patchgan_amd.infer:patchgan_infer']},
    install_requires=['torch', 'numpy', 'tqdm', 'pyyaml', 'pillow'],
)
