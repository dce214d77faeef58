"""Where a kernel's scratch traffic (register spills, local arrays the optimiser could not keep in registers) sits: per loop of the
annotated ISA (header label, nesting depth) the number of scratch loads / stores, barriers and other instructions.
    cd /tmp && hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S $REPO/neutfem_amd/csrc/neutfem_hip.hip -o nf.s
    python profiles/tools/spill_map.py nf.s [substring of the mangled kernel name] | c++filt
In k_resident_keff the CG loop is the depth-3 loop with four barriers (line sweep | p.q | |r|^2 | new p); its depth-4 child decodes
lane slots beyond the first 512 (never entered by the BASELINE configurations)."""
import re,sys
s=open(sys.argv[1]).read(); want=sys.argv[2] if len(sys.argv)>2 else ""
for m in re.finditer(r'^(_ZN2nf\S+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S|re.M):
    name,body=m.group(1),m.group(2).splitlines()
    if want not in name: continue
    fs=re.search(r'\.amdhsa_kernel '+re.escape(name)+r'.*?private_segment_fixed_size (\d+).*?next_free_vgpr (\d+)',s,re.S)
    print(name, 'frame',fs.group(1),'B/lane, vgpr',fs.group(2))
    # loop nest: a block belongs to the innermost loop named by its 'in Loop: Header=' / 'This Loop Header' comment
    parent={}; cur=('-',0); stats={}
    for i,l in enumerate(body):
        if re.match(r'^\.LBB\d+_\d+:',l) or re.match(r'^; %bb',l):
            lab=re.match(r'^(\.LBB\d+_\d+)',l); hdr=None; d=0
            for k in range(i,min(i+14,len(body))):
                t=body[k]
                if k>i and not (t.lstrip().startswith(';') ): break
                mm=re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)',t)
                if mm: hdr,d=mm.group(1),int(mm.group(2))
                mm=re.search(r'This (?:Inner )?Loop Header: Depth=(\d+)',t)
                if mm and lab: hdr,d=lab.group(1)[2:],int(mm.group(1))
            cur=(hdr or '-',d)
        st=stats.setdefault(cur,[0,0,0,0])
        if 'scratch_load' in l: st[0]+=1
        elif 'scratch_store' in l: st[1]+=1
        elif 's_barrier' in l: st[2]+=1
        elif l.startswith('\t') and not l.startswith('\t.') and not l.lstrip().startswith(';'): st[3]+=1
    for (h,d),(ld,stt,b,n) in sorted(stats.items(), key=lambda kv:(kv[0][1],kv[0][0])):
        if ld or stt or b: print(f"  depth {d} loop {h:12s} scratch loads {ld:3d} stores {stt:3d} barriers {b:2d} other instructions {n}")
