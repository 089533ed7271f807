"""Packaging of the MI355X style-transfer build: `pip install style-transfer-pytorch_amd` (or `-e`) after
`python style-transfer-pytorch_amd/build.py` has produced lib/libst_amd.so (hipcc, gfx950).  Same distribution
surface as the reference's setup.py:13-16: package `style_transfer`, console script `style_transfer`.

The HIP library is shipped INSIDE the package (style_transfer/lib/libst_amd.so, copied by the build step below), so a
non-editable install finds it too; an editable install / repo checkout uses lib/ next to the package (_hip.LIB_PATH)."""
import os
import shutil

import setuptools
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class build_py_with_library(build_py):
    def run(self):
        super().run()
        src = os.path.join(HERE, 'lib', 'libst_amd.so')
        if not os.path.exists(src):
            raise RuntimeError('lib/libst_amd.so is missing: run `python style-transfer-pytorch_amd/build.py` first '
                               '(hipcc, gfx950); the package has no CPU fallback')
        dst = os.path.join(self.build_lib, 'style_transfer', 'lib')
        os.makedirs(dst, exist_ok=True)
        shutil.copy2(src, os.path.join(dst, 'libst_amd.so'))


setuptools.setup(
    name='style-transfer-pytorch-amd',
    version='0.3',
    description='Neural style transfer with a hand-written HIP hot path for AMD MI355X (gfx950).',
    packages=['style_transfer'],
    package_dir={'style_transfer': 'style_transfer'},
    package_data={'style_transfer': ['lib/*.so']},
    cmdclass={'build_py': build_py_with_library},
    entry_points={'console_scripts': ['style_transfer=style_transfer.cli:main']},
    install_requires=['numpy', 'Pillow', 'torch', 'tqdm'],
    extras_require={'web': ['aiohttp>=3.7.2']},
    python_requires='>=3.8',
)
