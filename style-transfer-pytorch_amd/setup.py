"""Packaging of the MI355X style-transfer build: `pip install -e style-transfer-pytorch_amd` after
`python style-transfer-pytorch_amd/build.py` has produced lib/libst_amd.so (hipcc, gfx950).  Same distribution
surface as the reference's setup.py:13-16: package `style_transfer`, console script `style_transfer`."""
import setuptools

setuptools.setup(
    name='style-transfer-pytorch-amd',
    version='0.2',
    description='Neural style transfer with a hand-written HIP hot path for AMD MI355X (gfx950).',
    packages=['style_transfer'],
    package_dir={'style_transfer': 'style_transfer'},
    entry_points={'console_scripts': ['style_transfer=style_transfer.cli:main']},
    install_requires=['numpy', 'Pillow', 'torch', 'tqdm'],
    extras_require={'web': ['aiohttp>=3.7.2']},
    python_requires='>=3.8',
)
