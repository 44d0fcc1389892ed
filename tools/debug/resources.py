"""Per-kernel register / scratch summary from the build's resource report (csrc/liblcf_hip.resources.txt).
Usage: python tools/debug/resources.py [regex on the demangled kernel name]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
txt = open(os.path.join(ROOT, 'lightcurve_fitting_amd', 'csrc', 'liblcf_hip.resources.txt')).read()
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else '.')
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    dn = subprocess.run(['c++filt', '-p', name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace('(anonymous namespace)::', '')
    if not pat.search(dn):
        continue
    f = {k: re.search(k + r'[^:]*: (\d+)', b).group(1) for k in (' VGPRs', 'ScratchSize', 'SGPRs Spill', 'VGPRs Spill',
                                                               'Occupancy')}
    print(f'{dn[:44]:44s} VGPR {f[" VGPRs"]:>4} scratch {f["ScratchSize"]:>4} sgpr-spill {f["SGPRs Spill"]:>4} '
          f'vgpr-spill {f["VGPRs Spill"]:>3} occupancy {f["Occupancy"]}')
