"""Per-kernel resources of libcrfconv_amd.so from the code object's metadata note: VGPRs, AGPRs, SGPRs, LDS bytes, private-segment
(scratch) bytes.  usage: python3 scratch/kernel_resources.py [lib.so] [name filter]"""
import os, re, subprocess, sys, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.so') else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'crfconv_amd', 'libcrfconv_amd.so')
flt = [a for a in sys.argv[1:] if not a.endswith('.so')]
LLVM = '/opt/rocm/lib/llvm/bin'


def _notes_of(path, td, tag):
    fat, co = '%s/%s.fat' % (td, tag), '%s/%s.co' % (td, tag)
    if subprocess.run([LLVM + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fat, path], capture_output=True).returncode:
        return ''                                    # a unit without device code
    subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                    '--output=' + co], check=True)
    return subprocess.run([LLVM + '/llvm-readelf', '--notes', co], check=True, capture_output=True, text=True).stdout


def kernels(lib):
    """One code object per translation unit: read them from the objects next to the library (csrc/build/*.o)."""
    import glob
    objs = sorted(glob.glob(os.path.join(os.path.dirname(lib), 'csrc', 'build', '*.o')))
    out = []
    with tempfile.TemporaryDirectory() as td:
        for n, o in enumerate(objs):
            txt = _notes_of(o, td, 'u%d' % n)
            for blk in re.split(r'\n\s+- \.agpr_count:', txt)[1:]:
                blk = '.agpr_count:' + blk
                g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk)
                out.append(dict(name=g('name').group(1), vgpr=int(g('vgpr_count').group(1)), agpr=int(g('agpr_count').group(1)),
                                sgpr=int(g('sgpr_count').group(1)), lds=int(g('group_segment_fixed_size').group(1)),
                                scratch=int(g('private_segment_fixed_size').group(1)), unit=os.path.basename(o)))
    return out


if __name__ == '__main__':
    ks = kernels(lib)
    dem = subprocess.run(['c++filt'], input='\n'.join(k['name'] for k in ks), capture_output=True, text=True).stdout.split('\n')
    for k, d in zip(ks, dem):
        if flt and not any(f in d for f in flt):
            continue
        print('%-110s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %d' % (re.sub(r'\(.*', '', d)[:110], k['vgpr'], k['agpr'], k['sgpr'], k['lds'], k['scratch']))
    print('%d kernels, %d with scratch' % (len(ks), sum(1 for k in ks if k['scratch'])))
