"""One-off: crfconv_amd/ops.py -> the package crfconv_amd/ops/ (one module per operator family).  Cross-module names are imported at
the BOTTOM of each module (every reference is inside a function body), so import cycles between the families are harmless."""
import ast, builtins, os, re, symtable, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, 'crfconv_amd', 'ops.py')).read().split('\n')
L = lambda a, b: src[a - 1:b]          # 1-based inclusive line ranges


def find(prefix, start=1):
    for i in range(start - 1, len(src)):
        if src[i].startswith(prefix):
            return i + 1
    raise KeyError(prefix)


sec = {k: find('# ------------------------------------------------------------------------------ ' + k) for k in
       ('CRF mean field', 'discrete (label-space)', 'deferred weight gradients', 'per-point Linear', 'BatchNorm step counters',
        'BatchNorm (+ LeakyReLU)', 'Linear -> BatchNorm -> LeakyReLU', 'residual join', 'gather / max-pool', 'training loss', 'PointConv')}
end = find('__all__ = ')
ptr_array = (find('def _ptr_array'), find('class _CrfMatricesBatched') - 1)
tick_a = find('_TICKETS = {}')
tick_b = find('class _MeanFieldWide') - 1
gs_a = find('_sync_ws = {}')
gs_b = find('def _mlp_small_ok') - 1            # _sync_ws, gridsync_ws, _small_mlp_disabled
fw_a = find('def fail_word_ptrs')
fw_b = find('class _MLPSmall(') - 1

STATE = '''

class _State:
    """Switches the tests and bench.py's experiment knobs flip at run time (one object: the operator modules read it, a caller sets
    ``ops.state.<name>``)."""
    no_join = False              # tests: True = lin_out, bn_apply and add_lrelu as separate nodes (the fused nodes must give the same results)
    no_fork = False              # tests: True = autograd's own accumulation pass instead of the fork chain
    no_prefold = False           # tests: True = one fold launch inside every PointConv layer
    # set by check_gridsync after a barrier failure (CRFCONV_NO_ONE_LAUNCH_MLP: from the start): launch-separated forward from then on
    small_mlp_disabled = __import__('os').environ.get('CRFCONV_NO_ONE_LAUNCH_MLP') is not None
    # below this many rows the tiled product (gemm.hip) and the small-MLP nodes; swept on the step: 4096 -> 4.733 ms, 12288 (the
    # 10 240-row level joins the small forms) -> 4.694 ms, 65536 -> 4.736 ms
    mfma_min_rows = 12288


state = _State()
'''

mods = {}
mods['_base'] = (['"""Shared pieces of the operator modules: conversions, per-stream zero words, grid-barrier workspace and its failure check, run-time switches."""'],
                 L(6, sec['CRF mean field'] - 1) + L(*ptr_array) + [''] + L(tick_a, tick_b) + L(gs_a, gs_b) + L(fw_a, fw_b) + STATE.split('\n'))
crf_body = L(sec['CRF mean field'], ptr_array[0] - 1) + L(ptr_array[1] + 1, tick_a - 1) + L(tick_b + 1, sec['deferred weight gradients'] - 1)
mods['crf'] = (['"""Continuous CRF mean field (dense and wide), its matrices and their riders, the discrete (label-space) CRF layer."""'], crf_body)
mods['defer'] = (['"""Deferred parameter work of a backward pass: queues and the batched launches at its end."""'],
                 L(sec['deferred weight gradients'], sec['per-point Linear'] - 1))
mods['dense'] = (['"""Per-point Linear, BatchNorm step counters, BatchNorm (+ LeakyReLU)."""'],
                  L(sec['per-point Linear'], sec['Linear -> BatchNorm -> LeakyReLU'] - 1))
mlp_body = L(sec['Linear -> BatchNorm -> LeakyReLU'], gs_a - 1) + L(gs_b + 1, fw_a - 1) + L(fw_b + 1, sec['residual join'] - 1)
mods['mlp'] = (['"""Linear -> BatchNorm -> LeakyReLU blocks as single nodes: row-streaming forms, the classifier head, coarse-level forms and groups."""'], mlp_body)
mods['rows'] = (['"""Residual join, LeakyReLU, row gather, neighbour max-pool."""'], L(sec['residual join'], sec['training loss'] - 1))
mods['loss'] = (['"""Weighted soft-max cross-entropy (trainval.py:101-104)."""'], L(sec['training loss'], sec['PointConv'] - 1))
mods['pointconv'] = (['"""PointConv: rel-pos moments, the layer node, the batched BatchNorm-1 prefold."""'], L(sec['PointConv'], end - 1))

REN = [(r'\b_NO_JOIN_ENV\b', 'state.no_join'), (r'\b_NO_FORK_ENV\b', 'state.no_fork'), (r'\b_NO_PREFOLD_ENV\b', 'state.no_prefold'),
       (r'\b_small_mlp_disabled\b', 'state.small_mlp_disabled'), (r'\b_MFMA_MIN_ROWS\b', 'state.mfma_min_rows')]
DROP = [r'^state\.no_join = False', r'^state\.no_fork = False', r'^state\.no_prefold = False', r'^state\.small_mlp_disabled = __import__', r'^state\.mfma_min_rows = 12288',
        r'^\s*global state\.small_mlp_disabled']

texts = {}
for name, (doc, body) in mods.items():
    t = '\n'.join(body)
    if name != '_base':
        for a, b in REN:
            t = re.sub(a, b, t)
    else:
        for a, b in REN:
            t = re.sub(a, b, t)
    t = '\n'.join(l for l in t.split('\n') if not any(re.search(d, l) for d in DROP))
    texts[name] = t

HEAD = 'import ctypes\n\nimport torch\n\nfrom .. import _lib\nfrom ..graph import NeighborTable, ptr, require_gpu, stream_ptr\n'
defined = {}
for name, t in texts.items():
    full = (HEAD if name != '_base' else '') + t
    if name == '_base':
        full = full.replace('from . import _lib', 'from .. import _lib').replace('from .graph import', 'from ..graph import')
    tree = ast.parse(full)
    d = set()
    for n in tree.body:
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            d.add(n.name)
        elif isinstance(n, ast.Assign):
            for tg in n.targets:
                if isinstance(tg, ast.Name):
                    d.add(tg.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                d.add((a.asname or a.name).split('.')[0])
    defined[name] = d
    texts[name] = full


def globals_used(code, fname):
    used = set()

    def walk(tb):
        for s in tb.get_symbols():
            if s.is_global() or (tb.get_type() == 'module' and s.is_referenced()):
                used.add(s.get_name())
        for c in tb.get_children():
            walk(c)
    walk(symtable.symtable(code, fname, 'exec'))
    return used


out_dir = os.path.join(ROOT, 'crfconv_amd', 'ops')
os.makedirs(out_dir, exist_ok=True)
order = ['_base', 'defer', 'crf', 'dense', 'mlp', 'rows', 'loss', 'pointconv']
for name in order:
    t = texts[name]
    need = globals_used(t, name) - defined[name] - set(dir(builtins))
    imports = {}
    for n in sorted(need):
        owners = [m for m in order if m != name and n in defined[m] and n not in ('ctypes', 'torch', '_lib', 'NeighborTable', 'ptr', 'require_gpu', 'stream_ptr')]
        if not owners:
            print('UNRESOLVED in %s: %s' % (name, n))
            continue
        imports.setdefault(owners[0], []).append(n)
    top = ''
    bottom = ''
    for m, names in imports.items():
        line = 'from .%s import %s' % (m, ', '.join(names))
        line = '\n'.join(__import__('textwrap').wrap(line, 150, subsequent_indent='    ', break_long_words=False)) if len(line) > 150 else line
        if len(line) > 150 or '\n' in line:
            line = 'from .%s import (%s)' % (m, ', '.join(names))
            line = '\n'.join(__import__('textwrap').wrap(line, 150, subsequent_indent='    ', break_long_words=False))
        if m == '_base':
            top += line + '\n'
        else:
            bottom += line + '  # noqa: E402\n' if '\n' not in line else line + '  # noqa: E402\n'
    doc = mods[name][0][0]
    if name == '_base':
        body = t
        text = doc + '\n' + body.rstrip('\n') + '\n'
    else:
        body = t[len(HEAD):]
        text = doc + '\n' + HEAD + top + '\n' + body.strip('\n') + '\n'
        if bottom:
            text += '\n\n# names of the sibling modules, imported LAST: every use is inside a function body, so import cycles between the families are harmless\n' + bottom
    open(os.path.join(out_dir, name + '.py'), 'w').write(text)
    print(name, len(text.split('\n')), 'lines; imports', {m: len(v) for m, v in imports.items()})
