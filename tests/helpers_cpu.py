"""fixture unpacking without torch/GPU dependencies"""


def sites_from(g):
    return {tuple(int(v) for v in k.split('_')[1:]): g[k] for k in g.files if k.startswith('site_')}


def env_from(g, prefix):
    C, T = {}, {}
    for k in g.files:
        if k.startswith(prefix + 'C_') or k.startswith(prefix + 'T_'):
            x, y, vx, vy = (int(v) for v in k[len(prefix) + 2:].split('_'))
            (C if k[len(prefix)] == 'C' else T)[((x, y), (vx, vy))] = g[k]
    return C, T
