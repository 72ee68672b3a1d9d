cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(p)
for k in dir(p):
    if not k.startswith('_'):
        try: print(k, getattr(p,k))
        except Exception as e: pass
PY
/opt/rocm/bin/rocminfo | grep -i "lds\|Workgroup Max\|Wavefront\|Max Waves\|Compute Unit\|Local Mem\|GROUP" | head -30
