function varargout=dbat_hip_mex(varargin) %#ok<STOUT,INUSD>
%DBAT_HIP_MEX Stub for the MEX gateway of the MI355X bundle core.
%   Shadowed by the compiled MEX file when present (same convention as
%   code/test/postcov/icpc_mex.m in DBAT).
error('DBAT:dbat_hip_mex:notCompiled', ...
      'MEX file not found. Compile mex/dbat_hip_mex.cpp against libdbat_hip.so.');
