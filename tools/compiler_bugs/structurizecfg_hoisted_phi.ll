; Reduced from path_trace_kernel<0,false,false,true,true> (goblin_amd, trace.h trav_other with the tie rule inlined).
; llc -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -O3 red4.ll -print-after=structurizecfg
; Lanes that take check -> tie -> tiedone -> accept must leave with %acc = (inst, cand); after StructurizeCFG they leave with %rej.
target triple = "amdgcn-amd-amdhsa"
declare i32 @llvm.amdgcn.workitem.id.x()

define amdgpu_kernel void @f(ptr addrspace(1) %in, ptr addrspace(1) %out, ptr addrspace(1) %ord, i32 %inst, i32 %n) {
entry:
  %tid = call i32 @llvm.amdgcn.workitem.id.x()
  %gp = getelementptr <4 x i32>, ptr addrspace(1) %in, i32 %tid
  %v = load <4 x i32>, ptr addrspace(1) %gp
  %hit_inst = extractelement <4 x i32> %v, i64 0
  %hit_tri = extractelement <4 x i32> %v, i64 1
  %h0 = insertelement <2 x i32> poison, i32 %hit_inst, i64 0
  %hit0 = insertelement <2 x i32> %h0, i32 %hit_tri, i64 1
  %base = insertelement <2 x i32> poison, i32 %inst, i64 0
  br label %loop
loop:
  %i = phi i32 [ 0, %entry ], [ %i1, %join ]
  %hit = phi <2 x i32> [ %hit0, %entry ], [ %r, %join ]
  %ht = phi float [ 0x7FF0000000000000, %entry ], [ %rt, %join ]
  %k = add i32 %i, %tid
  %cp = getelementptr i32, ptr addrspace(1) %out, i32 %k
  %cand = load i32, ptr addrspace(1) %cp
  %tp = getelementptr float, ptr addrspace(1) %ord, i32 %k
  %t = load float, ptr addrspace(1) %tp
  %c_pass = fcmp ole float %t, %ht
  br i1 %c_pass, label %check, label %join
check:
  %teq = fcmp une float %t, %ht
  %hi = extractelement <2 x i32> %hit, i64 0
  %ine = icmp ne i32 %hi, %inst
  %c_nontie = select i1 %teq, i1 true, i1 %ine
  br i1 %c_nontie, label %accept, label %tie
tie:
  %cur = extractelement <2 x i32> %hit, i64 1
  %ap = getelementptr i32, ptr addrspace(1) %in, i32 %cur
  %a = load i32, ptr addrspace(1) %ap
  %x = xor i32 %a, %cand
  %xz = icmp eq i32 %x, 0
  br i1 %xz, label %same, label %diff
same:
  %s1 = lshr i32 %a, 8
  %sr = icmp ugt i32 %cand, %s1
  br label %tiedone
diff:
  %d1 = and i32 %x, 16
  %dr = icmp ne i32 %d1, 0
  br label %tiedone
tiedone:
  %c_rule = phi i1 [ %sr, %same ], [ %dr, %diff ]
  %rej = shufflevector <2 x i32> %base, <2 x i32> %hit, <2 x i32> <i32 0, i32 3>
  br i1 %c_rule, label %accept, label %join
accept:
  %acc = insertelement <2 x i32> %base, i32 %cand, i64 1
  br label %join
join:
  %r = phi <2 x i32> [ %hit, %loop ], [ %acc, %accept ], [ %rej, %tiedone ]
  %rt = phi float [ %ht, %loop ], [ %t, %accept ], [ %ht, %tiedone ]
  %i1 = add i32 %i, 1
  %done = icmp eq i32 %i1, %n
  br i1 %done, label %exit, label %loop
exit:
  %op = getelementptr <2 x i32>, ptr addrspace(1) %out, i32 %tid
  store <2 x i32> %r, ptr addrspace(1) %op
  %op2 = getelementptr float, ptr addrspace(1) %ord, i32 %tid
  store float %rt, ptr addrspace(1) %op2
  ret void
}
