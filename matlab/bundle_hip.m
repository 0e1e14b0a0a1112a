function [s,ok,iters,s0,E]=bundle_hip(s,varargin)
%BUNDLE_HIP Drop-in for BUNDLE that runs the adjustment on an MI355X.
%
%   [S,OK,N,S0,E]=BUNDLE_HIP(S[,N][,DAMP][,'trace'][,TOL][,'absterm']
%   [,'singulartest'|'nosingulartest'][,'pmdof'][,'dofverb']) takes the same
%   arguments as BUNDLE and returns the same values.  The residual/Jacobian model
%   (BROWN_EULER_CAM4), the normal equations and the damping loops of
%   BUNDLE/LSA run in libdbat_hip.so via DBAT_HIP_MEX.

% --- argument conventions of bundle.m:78-132
maxIter=20; damping='gna'; singularTest=true; absTerm=false; convTol=1e-6;
doTrace=false; pmDof=false; dofVerb=false;
while ~isempty(varargin)
    a=varargin{1}; varargin(1)=[];
    if isnumeric(a) && isscalar(a)
        if a==round(a), maxIter=a; else, convTol=a; end
    elseif ischar(a)
        switch lower(a)
          case {'none','gm','gna','lm','lmp'}, damping=lower(a);
          case 'trace', doTrace=true;
          case 'singulartest', singularTest=true;
          case 'nosingulartest', singularTest=false;
          case 'absterm', absTerm=true;
          case 'pmdof', pmDof=true;
          case 'dofverb', dofVerb=true;
          otherwise, error('DBAT:bundle:badInput','Unknown damping');
        end
    elseif islogical(a)
        if a, error('DBAT:bundle:badInput','chirality veto is undefined (bundle.m:169)'); end
    else
        error('DBAT:bundle:badInput','Unknown parameter');
    end
end
% --- what BUNDLE has no argument for lives in the optional struct s.bundle.hip:
%   device        HIP device ordinal of this MATLAB worker (default 0)
%   shardRank, shardCount, commId   several workers, one GPU each: this worker's rank, the number of workers
%                 and the 128 bytes of dbat_hip_mex('commId') that rank 0 created and sent to the others
%   wantJ         true: E.final.weighted.J / E.final.unweighted.J as BUNDLE returns them (bundle.m:341-350;
%                 BUNDLE_COV 'CXX' / 'COPF' read them); default: only after a failed run, for the post-mortem
%   wantCov       false: no posterior covariance blocks in s.post.cov (default true)
%   deterministic true: fixed-order sums on the device (bit-identical runs)
%   termFun       @(Jp,r) -> logical: replaces the termination test of bundle.m:186-192 (default [])
%   vetoFun       @(x) -> logical: the veto the LSA functions call at each trial point (default []: none;
%                 bundle.m:169 only knows the undefined CHIRALITY)
hip=struct('device',0,'shardRank',0,'shardCount',1,'commId',uint8([]),'wantJ',false,'wantCov',true,...
           'deterministic',false,'termFun',[],'vetoFun',[]);
if isfield(s.bundle,'hip') && ~isempty(s.bundle.hip)
    fn=fieldnames(s.bundle.hip);
    for i=1:length(fn)
        if ~isfield(hip,fn{i}), error('DBAT:bundle:badInput','Unknown field s.bundle.hip.%s',fn{i}); end
        hip.(fn{i})=s.bundle.hip.(fn{i});
    end
end
% --- bundle.m:137-154
s.prior.IO.use(~s.bundle.est.IO)=false;
s.prior.EO.use(~s.bundle.est.EO)=false;
s.prior.OP.use(~s.bundle.est.OP)=false;
% --- flatten the struct (0-based indices, column-major, IP columns are
% image-major with ascending OP, prob2dbatstruct.m:343-365)
[pt,cam]=find(s.IP.vis);
z=@(a)zeroifnan(a);
P=struct('nImages',size(s.EO.val,2),'nOP',size(s.OP.val,2),'nIP',size(s.IP.val,2),...
         'distModel',unique(s.IO.model.distModel),'nK',s.IO.model.nK,'nP',s.IO.model.nP,...
         'ipCam',int32(cam-1),'ipPt',int32(pt-1),...
         'ipVal',s.IP.val(:,full(s.IP.ix(s.IP.vis))),'ipStd',s.IP.std(:,full(s.IP.ix(s.IP.vis))),...
         'IO',s.IO.val,'pxSize',s.IO.sensor.pxSize,'EO',s.EO.val(1:6,:),'OP',s.OP.val,...
         'estIO',uint8(s.bundle.est.IO),'estEO',uint8(s.bundle.est.EO(1:6,:)),...
         'estOP',uint8(s.bundle.est.OP),...
         'IOblock',int32(s.IO.struct.block),'EOblock',int32(s.EO.struct.block(1:6,:)),...
         'useIO',uint8(s.prior.IO.use),'priorIO',z(s.prior.IO.val),'stdIO',z(s.prior.IO.std),...
         'useEO',uint8(s.prior.EO.use(1:6,:)),'priorEO',z(s.prior.EO.val(1:6,:)),...
         'stdEO',z(s.prior.EO.std(1:6,:)),...
         'useOP',uint8(s.prior.OP.use),'priorOP',z(s.prior.OP.val),'stdOP',z(s.prior.OP.std));
if ~isscalar(P.distModel), error('Mixed lens distortion models not implemented.'); end
dampNo=find(strcmp(damping,{'gm','gna','lm','lmp'}))-1;
if strcmp(damping,'none'), dampNo=0; end
wantJ=2; if hip.wantJ, wantJ=1; end   % 2: the gateway ships J only after code -2 / -4
opt=struct('damping',dampNo,'maxIter',maxIter,'convTol',convTol,'absTerm',absTerm,...
           'singularTest',singularTest,'trace',true,'wantJ',wantJ,'wantCov',logical(hip.wantCov),...
           'liveTrace',doTrace,'deterministic',logical(hip.deterministic),'device',hip.device,...
           'shardRank',hip.shardRank,'shardCount',hip.shardCount,'commId',uint8(hip.commId),...
           'termFun',hip.termFun,'vetoFun',hip.vetoFun);
% (with 'trace' the gateway prints the LSA function's line per iteration while the loop runs: opt.liveTrace)
[x,code,iters,s0,res,damp,aux,T,ru,rw,time,CEOb,CIOu,COPb,Jw,Ju]=dbat_hip_mex(P,opt);
s0gw=s0;     % the gateway's sigma0 (dof = m-n): what its covariance blocks are scaled with
if doTrace, fprintf('%s: %d iterations, code %d.\n',mfilename,iters,code); end
% --- result packaging, bundle.m:341-358,449-491
if isempty(s.bundle.serial) || isempty(s.bundle.deserial), s=buildserialindices(s); end
E=struct('maxIter',maxIter,'convTol',convTol,'absTerm',absTerm,'singularTest',singularTest,...
         'chirality',false,'res',res,'trace',T,'time',time(1),'code',code,'usedIters',iters);
% where the time went on the device (hipEvent stage timers), bundle.m:287-294
E.timeStages=struct('linearise',time(2),'factorSolve',time(3),'backsub',time(4),'residual',time(5),'other',time(6));
switch damping
  case {'none','gm'}, E.damping=struct('name','gm');
  case 'gna', E.damping=struct('name','gna','alpha',damp,'mu',0.1,'alphaMin',1e-9);
  case 'lm',  E.damping=struct('name','lm','lambda',damp,'lambda0',damp(1),'lambdaMin',damp(1));
  case 'lmp'
    rho=aux(1:maxIter+2); step=aux(maxIter+3:end);
    E.damping=struct('name','lmp','delta',damp,'rho',rho(~isnan(rho)),...
                     'delta0',norm(serialize(s)),'rhoBad',0.25,'rhoGood',0.75,...
                     'step',step(~isnan(step)));
end
E.final=struct('unweighted',struct('r',ru),'weighted',struct('r',rw),'factorized',[]);
if ~isempty(Jw)   % on request, or after a failed run (bundle.m:341-350)
    E.final.weighted.J=Jw; E.final.unweighted.J=Ju;
end
ok=code==0;
if ok, s=deserialize(s,x); end
% --- bundle.m:358-365
aspect=ones(2,size(s.IO.val,2)); aspect(1,:)=1+s.IO.val(4,:);
s.post.sensor.imSize=s.IO.sensor.imSize;
s.post.sensor.pxSize=s.IO.sensor.pxSize.*aspect;
s.post.sensor.ssSize=s.post.sensor.imSize.*s.IO.sensor.pxSize.*aspect;
% --- residual scatter, bundle.m:449-464
s.post.res.IP=nan(size(s.IP.val)); s.post.res.IO=nan(size(s.IO.val));
s.post.res.EO=nan(size(s.EO.val)); s.post.res.OP=nan(size(s.OP.val));
s.post.res.IP(:)=ru(s.post.res.ix.IP);
ptCols=s.IP.ix(s.IP.vis);
s.post.res.IP=s.post.res.IP./s.IO.sensor.pxSize(:,s.IP.cam(ptCols));
s.post.res.IO(s.prior.IO.use)=ru(s.post.res.ix.IO);
s.post.res.EO(s.prior.EO.use)=ru(s.post.res.ix.EO);
s.post.res.OP(s.prior.OP.use)=ru(s.post.res.ix.OP);
% --- sigma0 = sqrt(r'r/(m+p-n)), bundle.m:466-491: with 'pmdof' the fixed coordinates of measured
% control points and the fixed elements of used camera stations count as observations (p)
if pmDof
    p=nnz(s.bundle.est.OP(:,any(s.IP.vis,2))==0)+nnz(s.bundle.est.EO(1:6,any(s.IP.vis,1))==0);
else
    p=0;
end
lenR=length(rw); lenX=length(x); dof=lenR+p-lenX;
if dofVerb, fprintf('%s: dof=%d+%d-%d=%d.\n',mfilename,lenR,p,lenX,dof); end
s0=sqrt((rw'*rw)/dof);     % (the gateway's own sigma0 is the p=0 value)
s.post.sigmas=s0*s.IP.sigmas;
E.numObs=lenR; E.numParams=lenX; E.redundancy=dof;
E.s0=s0; E.sigmas=s.post.sigmas;
% --- posterior covariance blocks from the device (bundle_cov.m:193-210 reads
% s.post.cov.CEO / COP when they are present)
if ok && hip.wantCov
    % the gateway scaled the blocks with ITS sigma0; with 'pmdof' the degrees of freedom differ (bundle.m:466-483), and
    % BUNDLE_COV scales with E.s0 (bundle_cov.m:63-70): same blocks, E.s0's scale
    if s0~=s0gw && isfinite(s0gw) && s0gw>0
        sc=(s0/s0gw)^2; CEOb=CEOb*sc; CIOu=CIOu*sc; COPb=COPb*sc;
    end
    % block diagonals from (i,j,v) triplets: no dense intermediate (roma: 78 963^2 doubles)
    m=size(s.EO.val,1); nI=size(s.EO.val,2);
    [bi,bj]=ndgrid(1:6,1:6); ofs=reshape(m*(0:nI-1),1,1,[]);
    s.post.cov.CEO=sparse(reshape(bi+ofs,[],1),reshape(bj+ofs,[],1),CEOb(:),m*nI,m*nI);
    nO=size(s.OP.val,2);
    [bi,bj]=ndgrid(1:3,1:3); ofs=reshape(3*(0:nO-1),1,1,[]);
    s.post.cov.COP=sparse(reshape(bi+ofs,[],1),reshape(bj+ofs,[],1),COPb(:),3*nO,3*nO);
    % IO: every array entry that maps to IO unknown k gets row/column k of CIOu,
    % per-image diagonal blocks only
    ix=zeros(size(s.IO.val)); ix(s.bundle.deserial.IO.dest)=s.bundle.deserial.IO.src;
    [r,cI]=find(ix); e=find(ix); ti=[]; tj=[]; tv=[];
    for c=unique(cI)'
        k=e(cI==c); [ki,kj]=ndgrid(k,k);
        ti=[ti;ki(:)]; tj=[tj;kj(:)]; tv=[tv;reshape(CIOu(ix(k),ix(k)),[],1)]; %#ok<AGROW>
    end
    s.post.cov.CIO=sparse(ti,tj,tv,numel(ix),numel(ix));
end
% --- bundle.m:368-446: parameter names and the post-mortem of a failed run, on the J the gateway
% shipped for it (same tools as BUNDLE: spnrank / eigs for code -2, dmperm for code -4)
[~,E.paramTypes]=serialize(s);     % bundle.m:162,368
E.weakness=struct('structural',[],'numerical',[]);
if code==-2 && ~isempty(Jw)
    Js=Jw*spdiags(1./sqrt(full(sum(Jw.^2,1)))',0,lenX,lenX);   % the scaled Jacobian of gauss_newton_armijo.m:166-170
    nr=nan; try, nr=spnrank(Js); catch, end
    E.weakness.numerical.rank=nr;
    E.weakness.numerical.deficiency=lenX-nr;
    if E.weakness.numerical.deficiency>0
        try
            shift=sqrt(eps);          % keeps eigs alive on an exactly singular matrix
            JTJ=Js'*Js+shift*speye(lenX);
            [V,D]=eigs(JTJ,E.weakness.numerical.deficiency,'SM',struct('issym',true,'isreal',true));
            d=diag(D)-shift;
            [~,i]=sort(abs(d),'ascend'); d=d(i); V=V(:,i);
            E.weakness.numerical.V=V; E.weakness.numerical.d=d;
            E.weakness.numerical.trace=trace(JTJ);
            E.weakness.numerical.suspectedParams=cell(1,size(V,2));
            for j=1:size(V,2)
                [~,k]=sort(abs(V(:,j)),'descend'); v=V(k,j);
                keep=abs(v)>mean([sqrt(1/size(V,1)),abs(v(1))]);   % halfway between average and largest
                E.weakness.numerical.suspectedParams{j}=struct('values',v(keep),'indices',k(keep),...
                                                               'params',{E.paramTypes(k(keep))});
            end
        catch
            E.weakness.numerical.suspectedParams={};
        end
    end
elseif code==-4 && ~isempty(Jw)
    dm=dmperm(Jw);
    E.weakness.structural=struct('dmperm',dm,'rank',nnz(dm),'deficiency',lenX-nnz(dm),...
                                 'suspectedParams',{E.paramTypes(dm==0)});
    E.weakness.numerical.rank=nan; E.weakness.numerical.deficiency=nan;
elseif code==-2 || code==-4
    % no J (several shards: dbat_hip_jacobian_csc is a one-rank call): nothing is claimed about the rank
    E.weakness.numerical.rank=nan; E.weakness.numerical.deficiency=nan;
else
    E.weakness.numerical.rank=lenX; E.weakness.numerical.deficiency=0;
end

function a=zeroifnan(a)
a(isnan(a))=0;
