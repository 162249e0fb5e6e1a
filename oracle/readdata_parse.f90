! readdata_parse.f90 -- TEST INFRASTRUCTURE (like everything under oracle/): the READ statement sequence of RADEX's readdata_
! [radex.so@0x1cf90-0x1e338; SURVEY.md App. A.2; the STOP texts are the binary's own, tests/golden/ref_lamda_corpus.json]
! restated in Fortran and run by a REAL Fortran runtime (flang's), so that the list-directed input semantics the two C / C++
! readers restate by hand -- items across records, exponent forms, strict integers, (i1,a) -- can be checked against a Fortran
! library instead of against each other.  The reference binary itself cannot be used for that part: its libgfortran calls are
! served by the loader's own shim (oracle/macho_ref.py).
!
!   readdata_parse FILE   ->  stdout: "OK" + the parsed tables (17 significant digits), or "STOP <text>", "IOERR <where>",
!                             "OOB <where>" (an index the reference would use outside its arrays)
program readdata_parse
  implicit none
  integer, parameter :: maxlev = 2999, maxline = 99999, maxpart = 9, maxcoll = 99999, maxtemp = 99
  character(len=1024) :: path
  character(len=120) :: specref, qnum, ptext
  double precision :: amass, e, g, a, f, eu, t(maxtemp), r(maxtemp)
  double precision, allocatable :: eterm(:), gstat(:)
  integer :: nlev, nline, npart, ncoll, ntemp, ios, i, k, dummy, iu, il, ip, id
  call get_command_argument(1, path)
  open(11, file=trim(path), status='old', iostat=ios)
  if (ios /= 0) call fail('IOERR open')
  read(11, *, iostat=ios)
  if (ios /= 0) call fail('IOERR header')
  read(11, '(a)', iostat=ios) specref
  if (ios /= 0) call fail('IOERR name')
  read(11, *, iostat=ios)
  if (ios /= 0) call fail('IOERR header')
  read(11, *, iostat=ios) amass
  if (ios /= 0) call fail('IOERR weight')
  read(11, *, iostat=ios)
  if (ios /= 0) call fail('IOERR header')
  read(11, *, iostat=ios) nlev
  if (ios /= 0) call fail('IOERR nlev')
  if (nlev < 1) call fail('STOP error: too few energy levels defined')
  if (nlev > maxlev) call fail('STOP error: too many energy levels defined')
  allocate(eterm(nlev), gstat(nlev))
  read(11, *, iostat=ios)
  if (ios /= 0) call fail('IOERR header')
  do i = 1, nlev
    read(11, *, iostat=ios) dummy, eterm(i), gstat(i), qnum
    if (ios /= 0) call fail('IOERR levels')
    if (dummy < 1 .or. dummy > nlev) call fail('STOP error:illegal level number')
  end do
  read(11, *, iostat=ios)
  if (ios /= 0) call fail('IOERR header')
  read(11, *, iostat=ios) nline
  if (ios /= 0) call fail('IOERR nline')
  if (nline < 1) call fail('STOP error: too few spectral lines defined')
  if (nline > maxline) call fail('STOP error: too many spectral lines defined')
  read(11, *, iostat=ios)
  if (ios /= 0) call fail('IOERR header')
  write(*, '(a)') 'OK'
  write(*, '(a,es24.16e3)') 'amass ', amass
  write(*, '(a,i0)') 'nlev ', nlev
  do i = 1, nlev
    write(*, '(a,2es25.16e3)') 'level ', eterm(i), gstat(i)
  end do
  write(*, '(a,i0)') 'nline ', nline
  do i = 1, nline
    read(11, *, iostat=ios) dummy, iu, il, a, f, eu
    if (ios /= 0) call fail('IOERR lines')
    if (dummy < 1 .or. dummy > nline) call fail('STOP error:illegal line number')
    if (iu < 1 .or. iu > nlev .or. il < 1 .or. il > nlev) call fail('OOB line level index')
    if (eterm(iu) - eterm(il) < 1d-30) call fail('STOP error:illegal line frequency')
    write(*, '(a,2i6,4es25.16e3)') 'line ', iu, il, a, f, eu, eterm(iu) - eterm(il)
  end do
  read(11, *, iostat=ios)
  if (ios /= 0) call fail('IOERR header')
  read(11, *, iostat=ios) npart
  if (ios /= 0) call fail('IOERR npart')
  if (npart < 1) call fail('STOP error: too few collision partners defined')
  if (npart > maxpart) call fail('STOP error: too many collision partners')
  write(*, '(a,i0)') 'npart ', npart
  do ip = 1, npart
    read(11, *, iostat=ios)
    if (ios /= 0) call fail('IOERR header')
    read(11, '(i1,a)', iostat=ios) id, ptext
    if (ios /= 0) call fail('IOERR partner id')
    if (id < 1 .or. id > 7) call fail('OOB partner id')
    read(11, *, iostat=ios)
    if (ios /= 0) call fail('IOERR header')
    read(11, *, iostat=ios) ncoll
    if (ios /= 0) call fail('IOERR ncoll')
    if (ncoll < 1) call fail('STOP error: too few collision rates defined')
    if (ncoll > maxcoll) call fail('STOP error: too many collision rates')
    read(11, *, iostat=ios)
    if (ios /= 0) call fail('IOERR header')
    read(11, *, iostat=ios) ntemp
    if (ios /= 0) call fail('IOERR ntemp')
    if (ntemp < 1) call fail('OOB ntemp')
    if (ntemp > maxtemp) call fail('STOP error: too many collision temperatures')
    read(11, *, iostat=ios)
    if (ios /= 0) call fail('IOERR header')
    read(11, *, iostat=ios) (t(k), k = 1, ntemp)
    if (ios /= 0) call fail('IOERR temps')
    write(*, '(a,3i8)') 'partner ', id, ncoll, ntemp
    write(*, '(a,99es25.16e3)') 'temps ', (t(k), k = 1, ntemp)
    read(11, *, iostat=ios)
    if (ios /= 0) call fail('IOERR header')
    do i = 1, ncoll
      read(11, *, iostat=ios) dummy, iu, il, (r(k), k = 1, ntemp)
      if (ios /= 0) call fail('IOERR rates')
      if (dummy < 1 .or. dummy > ncoll) call fail('STOP error:illegal collision number')
      if (iu < 1 .or. il < 1 .or. iu > maxlev .or. il > maxlev) call fail('OOB rate level index')
      if (iu > nlev .or. il > nlev) cycle
      write(*, '(a,2i6,99es25.16e3)') 'rate ', iu, il, (r(k), k = 1, ntemp)
    end do
  end do
  write(*, '(a)') 'END'
contains
  subroutine fail(what)
    character(len=*), intent(in) :: what
    write(*, '(a)') 'FAIL ' // what
    stop
  end subroutine
end program
